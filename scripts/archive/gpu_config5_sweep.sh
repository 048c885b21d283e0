#!/bin/bash
# config-5 workload (mixed terrain + odor + gait adhesion) per GPU at several shard sizes: the arithmetic behind sharding.shard_plan
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for n in 128 512 1024 2048 4096; do
  timeout 300 python bench.py --no-cpu-baseline --no-live-counters --terrain mixed --odor --cpg-adhesion 20 --worlds-per-gpu $n 2>/dev/null | grep '^{"metric"' | python -c "
import sys, json
d = json.loads(sys.stdin.read()); ms = d['roofline'].get('kernel_ms_per_launch') or d['roofline_physics']['kernel_ms_per_launch']
print('config5 worlds', $n, round(d['value'] / 1e6, 3), 'M  us/step', round(1e3 * ms / d['config']['steps_per_launch'], 2))"
done | tee gpurun_out/config5_sweep.log
