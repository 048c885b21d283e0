#!/bin/bash
# kernel A/B on the GPU box over prebuilt variants on a terrain workload: gpu_abt.sh <terrain> <variant>...
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T=$1; shift
for lib in "$@"; do
  export NMF_HIP_LIB=$PWD/build/libnmf_$lib.so
  timeout 300 python bench.py --no-cpu-baseline --no-live-counters --terrain $T 2>/dev/null | grep '^{"metric"' | python -c "
import sys, json
d = json.loads(sys.stdin.read()); c = d['config']
print('$lib', '$T', round(d['value'] / 1e6, 2), 'M', 'ms/launch', round(d['roofline']['kernel_ms_per_launch'], 3), 'contacts', round(c['mean_contacts'], 2), 'iters', round(c['mean_newton_iters'], 2), 'valid', d.get('valid'))"
done > gpurun_out/abt.log 2>&1
cat gpurun_out/abt.log
