#!/bin/bash
# the whole GPU suite, then the bench as the driver runs it (with other_configs) and at its default arguments
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -v Warn | tail -40 ) > gpurun_out/r6_pytest.log
( timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 ) > gpurun_out/r6_smoke.log
python bench.py --no-cpu-baseline --no-live-counters --steps 20 --warmup 5 > gpurun_out/r6_bench_driver.log 2>&1
python bench.py --no-cpu-baseline --no-live-counters --no-other-configs > gpurun_out/r6_bench_default.log 2>&1
tail -30 gpurun_out/r6_pytest.log; cat gpurun_out/r6_smoke.log
for f in gpurun_out/r6_bench_driver.log gpurun_out/r6_bench_default.log; do grep '^{"metric"' $f | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); c = d['config']
    print(round(d['value'] / 1e6, 2), 'M', 'ms/launch', round(d['roofline']['kernel_ms_per_launch'], 3), 'contacts', round(c['mean_contacts'], 2), 'iters', round(c['mean_newton_iters'], 2), 'valid', d.get('valid'), {k: round(v, 1) for k, v in c['solver_exits'].items() if k != 'unit' and v})
    for o in d.get('other_configs', []):
        print('   ', o.get('config')[:100], '|', round(o.get('value', 0) / 1e6, 3) if o.get('value') else o, 'M valid', o.get('valid'), 'ms', round(o.get('kernel_ms_per_launch', 0), 3), o.get('vision_kernel_ms_per_tick'), o.get('roofline_frac'))
"; done
