#!/bin/bash
# chunk-plan sweep for short launches (the driver's --steps 20) and the default 50
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
run() { timeout 200 python bench.py --no-cpu-baseline --no-live-counters "$@" 2>/dev/null | grep '^{"metric"' | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(round(d['value'] / 1e6, 2), 'M', round(d['roofline']['kernel_ms_per_launch'], 3), 'ms')"; }
{
for div in 2.0 2.5 3.0; do for mc in 3 4 5; do
  echo -n "steps 20 div $div max_chunks $mc: "
  NMF_CHUNK_DIV=$div NMF_MAX_CHUNKS=$mc run --steps 20 --warmup 5
done; done
for div in 2.0 2.5; do for mc in 5 8; do
  echo -n "steps 50 div $div max_chunks $mc: "
  NMF_CHUNK_DIV=$div NMF_MAX_CHUNKS=$mc run
done; done
} > gpurun_out/chunk_sweep.log 2>&1
cat gpurun_out/chunk_sweep.log
