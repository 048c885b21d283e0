#!/bin/bash
# Run on the GPU box (through gpurun).  rocprofv3 passes over one bench command:
#   1. --kernel-trace --stats            (durations)
#   2. --pmc FETCH_SIZE                  (HBM read bytes;  own pass, kernel-trace only)
#   3. --pmc WRITE_SIZE                  (HBM write bytes; own pass)
#   4-7. --pmc SQ_* issue / wait / LDS / lane-occupancy / instruction-cache counters (own passes, <= 8 counters each)
# usage: scripts/profile_bench.sh <tag> [bench args...]
set -u
TAG=${1:-r2}; shift || true
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-live-counters $*"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o "$TAG" -- $BENCH > "$OUT/bench_trace.log" 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o "$TAG" -- $BENCH > "$OUT/bench_fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o "$TAG" -- $BENCH > "$OUT/bench_write.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY --output-format csv -d "$OUT/pmc_sq" -o "$TAG" -- $BENCH > "$OUT/bench_sq.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_sq2" -o "$TAG" -- $BENCH > "$OUT/bench_sq2.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_BUSY_CU_CYCLES SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_MFMA_I8 --output-format csv -d "$OUT/pmc_sq3" -o "$TAG" -- $BENCH > "$OUT/bench_sq3.log" 2>&1
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_INSTS_BRANCH SQ_IFETCH SQ_WAIT_INST_ANY --output-format csv -d "$OUT/pmc_sq4" -o "$TAG" -- $BENCH > "$OUT/bench_sq4.log" 2>&1
find "$OUT" -name "*.csv" | wc -l
grep -h '"metric"' "$OUT"/bench_*.log | cut -c1-160
grep -il "error\|invalid\|unknown counter" "$OUT"/bench_sq3.log | head -2
