#!/bin/bash
# sweep of the chunk plan (NMF_CHUNK_DIV, NMF_MAX_CHUNKS) for 20- and 50-step launches; lib = build/libnmf_$1.so
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export NMF_HIP_LIB=$PWD/build/libnmf_$1.so
B="python bench.py --no-cpu-baseline --no-live-counters"
val() { grep '^{"metric"' | python -c "
import sys, json
for l in sys.stdin: print(round(json.loads(l)['value'] / 1e6, 2))"; }
{
for div in 1.4 1.6 2 2.5 3 4; do for mc in 3 4 5 8; do
  a=$(NMF_CHUNK_DIV=$div NMF_MAX_CHUNKS=$mc timeout 200 $B --steps 20 --warmup 5 2>/dev/null | val)
  echo "div $div max_chunks $mc : 20-step $a M"
done; done
} > gpurun_out/sweep.log 2>&1
cat gpurun_out/sweep.log
