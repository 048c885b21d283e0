#!/bin/bash
# Run on the GPU box (through gpurun): rocprofv3 kernel trace + stats of the default bench command.
# usage: scripts/profile_bench.sh <tag> [bench args...]
set -u
TAG=${1:-r1}; shift || true
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o "$TAG" -- \
  python "$GRAFT_REPO_ROOT/bench.py" --no-cpu-baseline "$@" > "$OUT/bench.log" 2>&1
ls -la "$OUT"
cat "$OUT"/*kernel_stats.csv
grep '"metric"' "$OUT/bench.log" | cut -c1-600
