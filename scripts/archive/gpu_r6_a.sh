#!/bin/bash
# round 6, first GPU pass: the new parity tests, the rewritten config-1 test, the noslip tests, then the bench at the driver's arguments
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -f gpurun_out/parity_ledger.jsonl
( timeout 1500 python -m pytest tests/test_hip_parity_r6.py "tests/test_hip_parity.py::test_single_world_simulation_mirrors_the_cpu_class" tests/test_hip_parity_r4.py tests/test_hip_parity_r5.py -m gpu -q -x -s 2>&1 | grep -v Warn | tail -60 ) > gpurun_out/r6a_pytest.log
python bench.py --no-cpu-baseline --no-live-counters --steps 20 --warmup 5 > gpurun_out/r6a_bench_driver.log 2>&1
tail -40 gpurun_out/r6a_pytest.log
grep '^{"metric"' gpurun_out/r6a_bench_driver.log | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); c = d['config']
    print(round(d['value'] / 1e6, 2), 'M', 'ms/launch', round(d['roofline']['kernel_ms_per_launch'], 3), 'contacts', round(c['mean_contacts'], 2), 'iters', round(c['mean_newton_iters'], 2), 'valid', d.get('valid'))
    for o in d.get('other_configs', []):
        print('   ', o.get('config')[:100], '|', round(o.get('value', 0) / 1e6, 3) if o.get('value') else o, 'M valid', o.get('valid'), 'ms', round(o.get('kernel_ms_per_launch', 0), 3), o.get('vision_kernel_ms_per_tick'), o.get('noslip_iterations'))
"
