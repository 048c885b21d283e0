#!/bin/bash
cd $GRAFT_REPO_ROOT
bash scripts/profile_bench.sh r3c > gpurun_out/prof_a.log 2>&1
bash scripts/profile_bench.sh r3c_s20 --steps 20 --warmup 5 > gpurun_out/prof_s20.log 2>&1
bash scripts/profile_vision_bench.sh r3_vision > gpurun_out/prof_v.log 2>&1
bash scripts/profile_bench.sh r3c_bio --joint-preset all_biological > gpurun_out/prof_bio.log 2>&1
bash scripts/gpu_dist_trace.sh r3_dist1 > /dev/null 2>&1
for f in gpurun_out/prof_a.log gpurun_out/prof_s20.log gpurun_out/prof_v.log gpurun_out/prof_bio.log; do tail -n 3 $f | cut -c1-160; done
