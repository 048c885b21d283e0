#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -30 ) > gpurun_out/c4_pytest.log
B="python bench.py --no-cpu-baseline"
line() { grep '^{"metric"' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); c = d['config']
    print(sys.argv[1], round(d['value'] / 1e6, 2), 'M', 'ms/launch', round(d['roofline']['kernel_ms_per_launch'], 3), 'contacts', round(c['mean_contacts'], 2), 'iters', round(c['mean_newton_iters'], 2), 'valid', d.get('valid'))
" "$1"; }
{
NMF_NO_CHUNKS=1 timeout 200 $B 2>/dev/null | line "diet kernel, no chunks (cpg)"
timeout 200 $B 2>/dev/null | line "diet kernel, sc1 chunks (cpg)"
timeout 200 $B --steps 20 --warmup 5 2>/dev/null | line "diet kernel, sc1 chunks (cpg, driver args)"
timeout 200 $B --workload replay 2>/dev/null | line "diet kernel, sc1 chunks (replay)"
timeout 200 $B --worlds-per-gpu 8192 2>/dev/null | line "diet kernel, sc1 chunks 8192 (cpg)"
( cd build/prev && timeout 200 python bench.py --no-cpu-baseline 2>/dev/null | line "prev commit kernel (no diet, no chunks)" )
echo "--- stage profile: prev commit"
( cd build/prev && timeout 200 python scripts/stage_profile.py 4096 2>/dev/null | tail -23 )
echo "--- stage profile: diet kernel"
NMF_NO_CHUNKS=1 timeout 200 python scripts/stage_profile.py 4096 2>/dev/null | tail -23
} > gpurun_out/c4_ab.log 2>&1
tail -6 gpurun_out/c4_pytest.log; cat gpurun_out/c4_ab.log
