#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
run() { timeout 200 python bench.py --no-cpu-baseline --no-live-counters "$@" 2>/dev/null | grep '^{"metric"' | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(round(d['value'] / 1e6, 2), 'M', round(d['roofline']['kernel_ms_per_launch'], 3), 'ms')"; }
{
for div in 1.7 2.0 2.4; do
  echo -n "steps 20 div $div: "; NMF_CHUNK_DIV=$div run --steps 20 --warmup 5
  echo -n "steps 50 div $div: "; NMF_CHUNK_DIV=$div run
done
} > gpurun_out/chunk_sweep3.log 2>&1
cat gpurun_out/chunk_sweep3.log
