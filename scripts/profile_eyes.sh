#!/bin/bash
# Run on the GPU box (through gpurun): rocprofv3 over BASELINE config 3 with rendered eyes (`bench.py --vision render`).
# 1. --kernel-trace --stats   2-3. --pmc SQ issue / lane / instruction-cache counters (separate passes)
# usage: scripts/profile_eyes.sh <tag> [bench args...]
set -u
TAG=${1:-r4_eyes}; shift || true
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CMD="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-live-counters --no-other-configs --vision render --steps 200 $*"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o "$TAG" -- $CMD > "$OUT/bench_trace.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY --output-format csv -d "$OUT/pmc_sq" -o "$TAG" -- $CMD > "$OUT/bench_sq.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_VMEM SQC_ICACHE_REQ SQC_ICACHE_HITS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d "$OUT/pmc_sq2" -o "$TAG" -- $CMD > "$OUT/bench_sq2.log" 2>&1
grep -h '"metric"' "$OUT"/bench_*.log | cut -c1-160
