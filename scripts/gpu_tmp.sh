cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( time timeout 900 python bench.py ) > gpurun_out/default_bench.log 2>&1
grep '^{"metric"' gpurun_out/default_bench.log | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print(round(d['value']/1e6,2), 'M valid', d['valid'], 'counters', r.get('counters'), 'traffic', r['traffic'], 'frac', round(r['frac'],4))
print('cpu_baseline', d.get('cpu_baseline'))
"; grep real gpurun_out/default_bench.log
( time timeout 900 python bench.py --steps 20 --warmup 5 ) > gpurun_out/driver_bench.log 2>&1
grep '^{"metric"' gpurun_out/driver_bench.log | cut -c1-200; grep real gpurun_out/driver_bench.log
