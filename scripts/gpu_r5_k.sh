#!/bin/bash
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
mkdir -p gpurun_out
python scripts/gpu_ab.py --bench="--steps 20 --warmup 5 --no-other-configs" --bench="--no-other-configs" --bench="--no-other-configs --obs-every 1" --bench="--no-other-configs --terrain blocks" tree sens nowarm > gpurun_out/r5_ab_k.txt 2>&1
cat gpurun_out/r5_ab_k.txt
