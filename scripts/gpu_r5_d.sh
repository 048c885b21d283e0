#!/bin/bash
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
mkdir -p gpurun_out
python scripts/gpu_ab.py --reps 2 --bench="--steps 20 --warmup 5 --no-other-configs" --bench="--no-other-configs" base bf nocount nofb base bf > gpurun_out/r5_ab_d.txt 2>&1
cat gpurun_out/r5_ab_d.txt
