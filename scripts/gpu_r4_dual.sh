#!/bin/bash
# round 4: correctness + A/B of a variant library (build/libnmf_<name>.so) — smoke, LEGS_ONLY parity tests, bench lines
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
V=${1:-dual}
export NMF_HIP_LIB=$PWD/build/libnmf_$V.so
{
echo "=== smoke ($V)"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
echo "=== tests"
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "single_step_parity or rollout_parity or full_size_batch or cpg_adhesion or single_world_and_launch or two_seconds" 2>&1 | grep -E "^E |passed|failed|Error" | head -20
} > gpurun_out/r4_dual_$V.log 2>&1
B="python bench.py --no-cpu-baseline --no-live-counters"
line() { grep '^{"metric"' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); c = d['config']
    print(sys.argv[1], round(d['value'] / 1e6, 2), 'M', 'ms/launch', round(d['roofline']['kernel_ms_per_launch'], 3), 'contacts', round(c['mean_contacts'], 2), 'iters', round(c['mean_newton_iters'], 2), 'valid', d.get('valid'))
" "$1"; }
{
timeout 200 $B 2>/dev/null | line "$V chunks (cpg)"
timeout 200 $B --steps 20 --warmup 5 2>/dev/null | line "$V chunks (cpg, driver args)"
} >> gpurun_out/r4_dual_$V.log 2>&1
cat gpurun_out/r4_dual_$V.log
