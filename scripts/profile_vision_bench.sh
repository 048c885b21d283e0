#!/bin/bash
# Run on the GPU box (through gpurun): rocprofv3 over BASELINE config 3 as the driver-visible bench runs it
# (`bench.py --vision resample`).  1. --kernel-trace --stats   2. --pmc FETCH_SIZE   3. --pmc WRITE_SIZE  (separate passes)
# usage: scripts/profile_vision_bench.sh <tag> [bench args...]
set -u
TAG=${1:-r3_vision}; shift || true
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CMD="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-live-counters --vision resample --steps 200 $*"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o "$TAG" -- $CMD > "$OUT/bench_trace.log" 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o "$TAG" -- $CMD > "$OUT/bench_fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o "$TAG" -- $CMD > "$OUT/bench_write.log" 2>&1
grep -h '"metric"' "$OUT"/bench_*.log | cut -c1-200
