#!/bin/bash
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
python scripts/gpu_ab.py --bench="--no-other-configs --joint-preset all_biological" --bench="--no-other-configs --joint-preset all_biological --terrain mixed" --bench="--no-other-configs --joint-preset all_possible" biol biow
NMF_HIP_LIB=$GRAFT_REPO_ROOT/build/libnmf_biow.so timeout 900 python -m pytest tests -m gpu -q -k "all_biological or ALL_BIOLOGICAL or all_possible or ALL_POSSIBLE or collapsing or other_skeletons" 2>&1 | grep -v Warn | tail -8
} > gpurun_out/r5_ab_l.txt 2>&1
cat gpurun_out/r5_ab_l.txt
