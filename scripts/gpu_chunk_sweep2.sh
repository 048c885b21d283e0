#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
run() { timeout 200 python bench.py --no-cpu-baseline --no-live-counters "$@" 2>/dev/null | grep '^{"metric"' | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(round(d['value'] / 1e6, 2), 'M', round(d['roofline']['kernel_ms_per_launch'], 3), 'ms')"; }
{
for mn in 1 2 3; do for mc in 4 5 6; do
  echo -n "steps 20 min_steps $mn max_chunks $mc: "
  NMF_MIN_CHUNK_STEPS=$mn NMF_MAX_CHUNKS=$mc run --steps 20 --warmup 5
done; done
for mn in 1 2; do
  echo -n "steps 50 min_steps $mn: "
  NMF_MIN_CHUNK_STEPS=$mn run
done
} > gpurun_out/chunk_sweep2.log 2>&1
cat gpurun_out/chunk_sweep2.log
