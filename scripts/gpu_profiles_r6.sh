#!/bin/bash
# round 5: the profiles of record — default and driver-argument bench under rocprofv3, the every-step observation ring, the eye
# renderer, stage cycles of the shipped kernel
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
bash scripts/profile_bench.sh r6 --no-other-configs > gpurun_out/prof_r6.log 2>&1
bash scripts/profile_bench.sh r6_s20 --no-other-configs --steps 20 --warmup 5 > gpurun_out/prof_r6_s20.log 2>&1
bash scripts/profile_bench.sh r6_obs1 --no-other-configs --obs-every 1 > gpurun_out/prof_r6_obs1.log 2>&1
bash scripts/profile_eyes.sh r6_eyes > gpurun_out/prof_r6_eyes.log 2>&1
python scripts/stage_profile.py --build -DNMF_TOPO_MASK=1 > /dev/null 2>&1
python scripts/stage_profile.py 1792 > gpurun_out/r6_stage_cycles.txt 2>&1
python scripts/stage_profile.py 1792 --terrain=blocks > gpurun_out/r6_terrain_stage_cycles.txt 2>&1
tail -3 gpurun_out/prof_r6.log | cut -c1-200; tail -3 gpurun_out/prof_r6_s20.log | cut -c1-200; tail -2 gpurun_out/prof_r6_obs1.log | cut -c1-200; tail -2 gpurun_out/prof_r6_eyes.log | cut -c1-200
grep -v Warn gpurun_out/r6_stage_cycles.txt | head -24
