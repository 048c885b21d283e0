#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
B="python bench.py --no-cpu-baseline --no-live-counters"
line() { grep '^{"metric"' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); c = d['config']
    print(sys.argv[1], round(d['value'] / 1e6, 2), 'M', 'ms/launch', round(d['roofline']['kernel_ms_per_launch'], 3))
" "$1"; }
{
for div in 1.5 2 2.5 3 4 7; do for mc in 6 8 12; do
  NMF_CHUNK_DIV=$div NMF_MAX_CHUNKS=$mc timeout 200 $B --steps 20 --warmup 5 2>/dev/null | line "20-step launches, div $div max_chunks $mc"
done; done
for div in 2 2.5 3 4 7; do for mc in 8 12; do
  NMF_CHUNK_DIV=$div NMF_MAX_CHUNKS=$mc timeout 200 $B 2>/dev/null | line "50-step launches, div $div max_chunks $mc"
done; done
} > gpurun_out/chunks.log 2>&1
cat gpurun_out/chunks.log
