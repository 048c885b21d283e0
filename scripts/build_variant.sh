#!/bin/bash
# Build a variant of libnmf_hip.so with extra -D flags for kernel A/B experiments (loaded with NMF_HIP_LIB=<path>).
# usage: scripts/build_variant.sh <name> [-DFLAG ...]   -> build/libnmf_<name>.so
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
mkdir -p "$ROOT/build"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -mllvm -amdgpu-sched-strategy=iterative-ilp \
  -fno-hip-fp32-correctly-rounded-divide-sqrt -freciprocal-math -fno-signed-zeros -fassociative-math -fno-trapping-math \
  -fno-math-errno -fapprox-func -mllvm -amdgpu-atomic-optimizer-strategy=None -fPIC -shared "$@" -I"$ROOT/include" -I"$ROOT/flygym_amd/csrc" \
  "$ROOT/flygym_amd/csrc/nmf_capi.hip" -o "$ROOT/build/libnmf_$NAME.so" 2>&1 | grep -v "occupancy target\|nmf_step_kernel(const\|\^\|warnings generated" || true
python "$ROOT/scripts/kernel_stats.py" "$ROOT/build/libnmf_$NAME.so" | grep "step_kernel" | head -4
