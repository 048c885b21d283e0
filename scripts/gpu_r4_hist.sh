#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python scripts/gpu_iter_hist.py > gpurun_out/r4_hist.log 2>&1
python bench.py --no-cpu-baseline --no-live-counters --steps 20 --warmup 5 > gpurun_out/r4_bench0.log 2>&1
tail -30 gpurun_out/r4_hist.log; tail -2 gpurun_out/r4_bench0.log
