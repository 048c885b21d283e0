#!/bin/bash
# Run on the GPU box: kernel-trace of the bench's distributed code path on ONE rank (NMF_BENCH_FORCE_DIST=1: process group,
# RCCL all-gather of the observation block per tick on RCCL's stream, double-buffered) — do the gather kernels of tick k
# overlap the persistent stepping launch of tick k + 1 or queue behind it?
set -u
TAG=${1:-r3_dist1}; shift || true
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export NMF_BENCH_FORCE_DIST=1 HSA_ENABLE_IPC_MODE_LEGACY=0
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o "$TAG" -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-live-counters --steps 500 $* > "$OUT/bench_trace.log" 2>&1
grep -h '"metric"' "$OUT"/bench_trace.log | cut -c1-200
tail -3 "$OUT"/bench_trace.log | cut -c1-200
