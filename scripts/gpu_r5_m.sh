#!/bin/bash
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
python scripts/stage_profile.py 1792 2>&1 | grep -v "Warn\|warn\|amdgpu" | grep "expansion\|final forces\|cycles per step"
python scripts/gpu_ab.py --bench="--steps 20 --warmup 5 --no-other-configs" --bench="--no-other-configs" --bench="--no-other-configs --joint-preset all_biological" exp exp2
NMF_HIP_LIB=$GRAFT_REPO_ROOT/build/libnmf_exp2.so timeout 900 python -m pytest tests/test_hip_parity_r4.py tests/test_hip_parity_r5.py tests/test_hip_parity.py -m gpu -q -x -k "contact_space or noslip_pass or every_step or single_step or solves_end and not blocks and not mixed" 2>&1 | grep -v Warn | tail -6
} > gpurun_out/r5_ab_m.txt 2>&1
cat gpurun_out/r5_ab_m.txt
