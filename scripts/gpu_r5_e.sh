#!/bin/bash
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
mkdir -p gpurun_out
python scripts/gpu_ab.py --bench="--steps 20 --warmup 5 --no-other-configs" --bench="--no-other-configs" bf bf2 > gpurun_out/r5_ab_e.txt 2>&1
python scripts/gpu_ab.py --bench="--steps 20 --warmup 5 --no-other-configs" --bench="--no-other-configs" --sweep NMF_CHUNK_DIV=1.4,1.5,1.7,1.8,2.0 bf2 >> gpurun_out/r5_ab_e.txt 2>&1
python scripts/gpu_ab.py --bench="--steps 20 --warmup 5 --no-other-configs" --bench="--no-other-configs" --sweep NMF_MAX_CHUNKS=4,6 bf2 >> gpurun_out/r5_ab_e.txt 2>&1
cat gpurun_out/r5_ab_e.txt
