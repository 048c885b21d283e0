#!/bin/bash
# usage: gpu_profile.sh <tag>   -> prof_<tag>_s20 (the driver's arguments) and prof_<tag> (default arguments)
cd $GRAFT_REPO_ROOT
TAG=${1:-r2b}
bash scripts/profile_bench.sh ${TAG}_s20 --steps 20 --warmup 5 > /dev/null
bash scripts/profile_bench.sh ${TAG} > /dev/null
grep -h '"metric"' gpurun_out/prof_${TAG}*/bench_trace.log | cut -c1-140
