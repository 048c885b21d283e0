#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -30 ) > gpurun_out/c3_pytest.log
B="python bench.py --no-cpu-baseline"
line() { grep '^{"metric"' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); c = d['config']
    print(sys.argv[1], round(d['value'] / 1e6, 2), 'M', 'ms/launch', round(d['roofline']['kernel_ms_per_launch'], 3), 'contacts', round(c['mean_contacts'], 2), 'iters', round(c['mean_newton_iters'], 2), 'valid', d.get('valid'))
" "$1"; }
{
python -c "
import torch, ctypes
from flygym_amd import HIPSimulation, make_model
fly, world, _ = make_model()
sim = HIPSimulation(world, n_worlds=64, device=0)
print('ok')
"
NMF_NO_CHUNKS=1 timeout 200 $B 2>/dev/null | line "diet kernel, no chunks (cpg)"
timeout 200 $B 2>/dev/null | line "diet kernel, chunks (cpg)"
timeout 200 $B --steps 20 --warmup 5 2>/dev/null | line "diet kernel, chunks (cpg, driver args)"
NMF_NO_CHUNKS=1 timeout 200 $B --steps 20 --warmup 5 2>/dev/null | line "diet kernel, no chunks (cpg, driver args)"
timeout 200 $B --workload replay 2>/dev/null | line "diet kernel, chunks (replay)"
export NMF_HIP_LIB=$PWD/build/libnmf_w3.so
NMF_NO_CHUNKS=1 timeout 200 $B 2>/dev/null | line "w3 kernel, no chunks (cpg)"
timeout 200 $B 2>/dev/null | line "w3 kernel, chunks (cpg)"
timeout 200 $B --steps 20 --warmup 5 2>/dev/null | line "w3 kernel, chunks (cpg, driver args)"
timeout 200 $B --workload replay 2>/dev/null | line "w3 kernel, chunks (replay)"
timeout 200 $B --worlds-per-gpu 3072 2>/dev/null | line "w3 kernel, 3072 worlds (one round)"
timeout 200 $B --worlds-per-gpu 6144 2>/dev/null | line "w3 kernel, chunks 6144 worlds"
timeout 200 $B --joint-preset legs_active_only 2>/dev/null | line "w3 active, chunks 4096"
unset NMF_HIP_LIB
timeout 200 $B --joint-preset legs_active_only 2>/dev/null | line "w2 active, chunks 4096"
} > gpurun_out/c3_ab.log 2>&1
tail -12 gpurun_out/c3_pytest.log; cat gpurun_out/c3_ab.log
