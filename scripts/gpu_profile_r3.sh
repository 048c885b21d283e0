#!/bin/bash
cd $GRAFT_REPO_ROOT
bash scripts/profile_bench.sh ${1:-r3a} > gpurun_out/prof_a.log 2>&1
bash scripts/profile_bench.sh ${1:-r3a}_s20 --steps 20 --warmup 5 > gpurun_out/prof_s20.log 2>&1
bash scripts/profile_vision_bench.sh r3_vision > gpurun_out/prof_v.log 2>&1
tail -3 gpurun_out/prof_a.log gpurun_out/prof_s20.log gpurun_out/prof_v.log | cut -c1-220
du -sh gpurun_out/prof_${1:-r3a} gpurun_out/prof_${1:-r3a}_s20 gpurun_out/prof_r3_vision
