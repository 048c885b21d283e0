#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -25 ) > gpurun_out/c2_pytest.log
B="python bench.py --no-cpu-baseline"
line() { grep '^{"metric"' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); c = d['config']
    print(sys.argv[1], round(d['value'] / 1e6, 2), 'M', 'ms/launch', round(d['roofline']['kernel_ms_per_launch'], 3), 'contacts', round(c['mean_contacts'], 2), 'iters', round(c['mean_newton_iters'], 2), 'valid', d.get('valid'))
" "$1"; }
{
( cd build/r1 && timeout 200 python bench.py --no-cpu-baseline 2>/dev/null | line "r1 kernel + r1 bench (cpg)" )
( cd build/r1 && timeout 200 python bench.py --no-cpu-baseline --workload replay 2>/dev/null | line "r1 kernel + r1 bench (replay)" )
NMF_BENCH_R1_PROTOCOL=1 timeout 200 $B 2>/dev/null | line "r2 kernel, r1 protocol (cpg)"
timeout 200 $B 2>/dev/null | line "r2 kernel, r2 protocol (cpg)"
timeout 200 $B --workload replay 2>/dev/null | line "r2 kernel (replay)"
for n in 2048 4096 8192; do timeout 200 $B --worlds-per-gpu $n 2>/dev/null | line "r2 cpg worlds $n"; done
for n in 2048 3072 4096 6144; do
  timeout 200 $B --joint-preset legs_active_only --worlds-per-gpu $n 2>/dev/null | line "active w2 worlds $n"
  NMF_HIP_LIB=$PWD/build/libnmf_w3.so timeout 200 $B --joint-preset legs_active_only --worlds-per-gpu $n 2>/dev/null | line "active w3 worlds $n"
done
} > gpurun_out/c2_ab.log 2>&1
tail -8 gpurun_out/c2_pytest.log; cat gpurun_out/c2_ab.log
