#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
B="python bench.py --no-cpu-baseline"
line() { grep '^{"metric"' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); c = d['config']
    print(sys.argv[1], round(d['value'] / 1e6, 2), 'M', 'ms/launch', round(d['roofline']['kernel_ms_per_launch'], 3))
" "$1"; }
{
for mc in 7 10 20; do for ms in 1 2; do
  NMF_MAX_CHUNKS=$mc NMF_MIN_CHUNK_STEPS=$ms timeout 200 $B --steps 20 --warmup 5 2>/dev/null | line "20-step launches, max_chunks $mc min_steps $ms"
done; done
for mc in 7 10 17 25; do
  NMF_MAX_CHUNKS=$mc NMF_MIN_CHUNK_STEPS=1 timeout 200 $B 2>/dev/null | line "50-step launches, max_chunks $mc"
done
NMF_MAX_CHUNKS=25 NMF_MIN_CHUNK_STEPS=1 timeout 200 $B --workload replay 2>/dev/null | line "50-step launches replay, max_chunks 25"
NMF_MAX_CHUNKS=10 NMF_MIN_CHUNK_STEPS=1 timeout 200 $B --workload replay 2>/dev/null | line "50-step launches replay, max_chunks 10"
} > gpurun_out/chunks.log 2>&1
cat gpurun_out/chunks.log
