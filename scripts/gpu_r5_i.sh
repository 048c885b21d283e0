#!/bin/bash
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
export NMF_HIP_LIB=$GRAFT_REPO_ROOT/build/libnmf_allp.so
mkdir -p gpurun_out
{
timeout 900 python -m pytest tests/test_hip_parity_r3.py tests/test_hip_parity.py tests/test_hip_parity_r5.py -m gpu -q -k "all_possible or ALL_POSSIBLE" 2>&1 | grep -v Warn | tail -15
python scripts/gpu_ab.py --bench="--no-other-configs --joint-preset all_possible" --bench="--no-other-configs --joint-preset all_possible --terrain mixed" allp
python scripts/gpu_exit_hist.py flat 1 100 all_possible 2>&1 | grep -v "Warn\|warn\|amdgpu"
} > gpurun_out/r5_allp.txt 2>&1
cat gpurun_out/r5_allp.txt
