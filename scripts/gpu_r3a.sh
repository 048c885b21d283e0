#!/bin/bash
# round 3, first GPU call: the whole gpu suite (new r3 tests included), smoke, and the schedule A/B of the stepping kernel
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests -m gpu -q --deselect tests/test_hip_parity_r3.py 2>&1 | tail -30 ) > gpurun_out/pytest_old.log
( timeout 1500 python -m pytest tests/test_hip_parity_r3.py -m gpu -q 2>&1 | tail -60 ) > gpurun_out/pytest_r3.log
( timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 ) > gpurun_out/smoke.log
B="python bench.py --no-cpu-baseline --no-live-counters"
line() { grep '^{"metric"' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); c = d['config']
    print(sys.argv[1], round(d['value'] / 1e6, 2), 'M', 'ms/launch', round(d['roofline']['kernel_ms_per_launch'], 3), 'contacts', round(c['mean_contacts'], 2), 'iters', round(c['mean_newton_iters'], 2), 'valid', d.get('valid'))
" "$1"; }
{
for sched in chunks paired; do
  export NMF_SCHED=$sched
  timeout 200 $B --steps 20 --warmup 5 2>/dev/null | line "$sched (cpg, driver args: 20-step launches)"
  timeout 200 $B 2>/dev/null | line "$sched (cpg, 50-step launches)"
  timeout 200 $B --steps-per-launch 250 2>/dev/null | line "$sched (cpg, 250-step launches)"
  timeout 200 $B --workload replay --steps 20 --warmup 5 2>/dev/null | line "$sched (replay, 20-step launches)"
  timeout 200 $B --workload replay 2>/dev/null | line "$sched (replay, 50-step launches)"
  timeout 200 $B --joint-preset all_biological --steps 20 --warmup 5 2>/dev/null | line "$sched (all_biological, 20-step launches)"
  timeout 200 $B --joint-preset all_biological 2>/dev/null | line "$sched (all_biological, 50-step launches)"
done
unset NMF_SCHED
timeout 200 $B --steps 20 --warmup 5 2>/dev/null | line "auto (cpg, driver args)"
} > gpurun_out/ab.log 2>&1
tail -5 gpurun_out/pytest_old.log; tail -40 gpurun_out/pytest_r3.log; cat gpurun_out/smoke.log; cat gpurun_out/ab.log
