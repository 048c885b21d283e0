#!/bin/bash
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
mkdir -p gpurun_out
python scripts/gpu_ab.py --bench="--steps 20 --warmup 5 --no-other-configs" --bench="--no-other-configs" --bench="--no-other-configs --terrain blocks" bf2 tree > gpurun_out/r5_ab_h.txt 2>&1
python scripts/gpu_ab.py --bench="--no-other-configs --joint-preset all_biological" --bench="--no-other-configs --joint-preset all_biological --terrain mixed" --bench="--no-other-configs --joint-preset all_possible" tree >> gpurun_out/r5_ab_h.txt 2>&1
cat gpurun_out/r5_ab_h.txt
