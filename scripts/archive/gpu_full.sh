#!/bin/bash
# the whole GPU suite + the bench at default and driver arguments, in-tree library
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 1700 python -m pytest tests -m gpu -q -x 2>&1 | tail -25 ) > gpurun_out/r4_pytest.log
python bench.py --no-cpu-baseline --no-live-counters > gpurun_out/r4_bench_default.log 2>&1
python bench.py --no-cpu-baseline --no-live-counters --steps 20 --warmup 5 > gpurun_out/r4_bench_driver.log 2>&1
tail -25 gpurun_out/r4_pytest.log
for f in gpurun_out/r4_bench_default.log gpurun_out/r4_bench_driver.log; do grep '^{"metric"' $f | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); c = d['config']
    print(round(d['value'] / 1e6, 2), 'M', 'ms/launch', round(d['roofline']['kernel_ms_per_launch'], 3), 'contacts', round(c['mean_contacts'], 2), 'iters', round(c['mean_newton_iters'], 2), 'valid', d.get('valid'))
"; done
