#!/bin/bash
# round 5: the whole GPU suite + the headline and a few workload lines on the in-tree library
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -60 ) > gpurun_out/r5_pytest.log
B="python bench.py --no-cpu-baseline --no-live-counters --no-other-configs"
line() { grep '^{"metric"' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); c = d['config']
    print(sys.argv[1], round(d['value'] / 1e6, 3), 'M', 'ms/launch', round(d['roofline']['kernel_ms_per_launch'], 3), 'contacts', round(c['mean_contacts'], 2), 'iters', round(c['mean_newton_iters'], 2), 'valid', d.get('valid'))
" "$1"; }
{
timeout 300 $B --steps 20 --warmup 5 2>/dev/null | line "cpg 20-step launches (driver args)"
timeout 300 $B 2>/dev/null | line "cpg default"
timeout 300 $B --worlds-per-gpu 1024 2>/dev/null | line "cpg 1024"
timeout 300 $B --worlds-per-gpu 2048 2>/dev/null | line "cpg 2048"
timeout 300 $B --terrain blocks 2>/dev/null | line "cpg terrain blocks"
timeout 300 $B --terrain mixed --odor --cpg-adhesion 20 --worlds-per-gpu 1024 2>/dev/null | line "config5 mixed+odor+adhesion 1024"
timeout 300 $B --joint-preset all_biological 2>/dev/null | line "all_biological"
timeout 300 $B --joint-preset legs_active_only 2>/dev/null | line "legs_active_only"
timeout 300 $B --workload replay 2>/dev/null | line "replay 4096"
} > gpurun_out/r5_workloads_a.log 2>&1
tail -40 gpurun_out/r5_pytest.log
cat gpurun_out/r5_workloads_a.log
