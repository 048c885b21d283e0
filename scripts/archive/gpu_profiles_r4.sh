#!/bin/bash
# round 4: the profiles of record — default and driver-argument bench under rocprofv3, stage cycles of the shipped kernel
cd $GRAFT_REPO_ROOT
bash scripts/profile_bench.sh r4 --no-other-configs > gpurun_out/prof_r4.log 2>&1
bash scripts/profile_bench.sh r4_s20 --no-other-configs --steps 20 --warmup 5 > gpurun_out/prof_r4_s20.log 2>&1
python scripts/stage_profile.py 1792 > gpurun_out/r4_stage_cycles.txt 2>&1
NMF_SOLVER=primal python scripts/stage_profile.py 1792 > gpurun_out/r4_stage_cycles_primal.txt 2>&1
NMF_SOLVER=nohist python scripts/stage_profile.py 1792 > gpurun_out/r4_stage_cycles_nohist.txt 2>&1
bash scripts/profile_bench.sh r4_bio --no-other-configs --joint-preset all_biological > gpurun_out/prof_r4_bio.log 2>&1
python scripts/stage_profile.py 1792 --joint-preset=all_biological > gpurun_out/r4_bio_stage_cycles.txt 2>&1
NMF_SOLVER=primal python scripts/stage_profile.py 1792 --joint-preset=all_biological > gpurun_out/r4_bio_stage_cycles_primal.txt 2>&1
tail -3 gpurun_out/prof_r4.log | cut -c1-200; tail -3 gpurun_out/prof_r4_s20.log | cut -c1-200; head -22 gpurun_out/r4_stage_cycles.txt | tail -19
