#!/bin/bash
# per-stage cycle breakdown of the shipped kernels (diagnostic build: libnmf_hip_prof.so must be current — `python scripts/stage_profile.py --build`)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
{
echo "== LEGS_ONLY, flat, replay protocol, 2048 worlds"; timeout 300 python scripts/stage_profile.py 2048
echo "== LEGS_ONLY, mixed terrain (Terrain<TP> kernel), 2048 worlds"; timeout 300 python scripts/stage_profile.py 2048 --terrain=mixed
echo "== ALL_BIOLOGICAL, flat, 1792 worlds (the diagnostic build holds 7 flies per CU: its stage accumulators take 224 B of LDS)"; timeout 300 python scripts/stage_profile.py 1792 --joint-preset=all_biological
} 2>/dev/null > gpurun_out/stage_r3.log
cat gpurun_out/stage_r3.log
