#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -60 ) > gpurun_out/r5_pytest.log
{ python scripts/gpu_exit_hist.py flat 1 400; python scripts/gpu_exit_hist.py blocks 1 300; python scripts/gpu_exit_hist.py mixed 20 300; python scripts/gpu_exit_hist.py flat 1 200 all_biological; } > gpurun_out/r5_exit_hist.txt 2>&1
B="python bench.py --no-cpu-baseline --no-live-counters --no-other-configs"
line() { grep '^{"metric"' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); c = d['config']
    print(sys.argv[1], round(d['value'] / 1e6, 3), 'M', 'ms/launch', round(d['roofline']['kernel_ms_per_launch'], 3), 'contacts', round(c['mean_contacts'], 2), 'iters', round(c['mean_newton_iters'], 2), 'valid', d.get('valid'), {k: round(v, 1) for k, v in c['solver_exits'].items() if k != 'unit' and v})
" "$1"; }
{
timeout 300 $B --steps 20 --warmup 5 2>/dev/null | line "cpg 20-step launches (driver args)"
timeout 300 $B 2>/dev/null | line "cpg default"
timeout 300 $B --terrain blocks 2>/dev/null | line "cpg terrain blocks"
timeout 300 $B --terrain mixed --odor --cpg-adhesion 20 --worlds-per-gpu 1024 2>/dev/null | line "config5 mixed+odor+adhesion 1024"
timeout 300 $B --joint-preset all_biological 2>/dev/null | line "all_biological"
} > gpurun_out/r5_workloads_b.log 2>&1
tail -40 gpurun_out/r5_pytest.log
cat gpurun_out/r5_exit_hist.txt gpurun_out/r5_workloads_b.log
