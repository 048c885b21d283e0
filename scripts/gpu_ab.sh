#!/bin/bash
# quick kernel A/B on the GPU box: bench lines for the in-tree library (and any NMF_HIP_LIB variants passed as args)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
B="python bench.py --no-cpu-baseline --no-live-counters"
line() { grep '^{"metric"' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); c = d['config']
    print(sys.argv[1], round(d['value'] / 1e6, 2), 'M', 'ms/launch', round(d['roofline']['kernel_ms_per_launch'], 3), 'contacts', round(c['mean_contacts'], 2), 'iters', round(c['mean_newton_iters'], 2), 'valid', d.get('valid'))
" "$1"; }
{
for lib in "" "$@"; do
  if [ -n "$lib" ]; then export NMF_HIP_LIB=$PWD/build/libnmf_$lib.so; fi
  tag=${lib:-tree}
  NMF_NO_CHUNKS=1 timeout 200 $B 2>/dev/null | line "$tag no chunks (cpg)"
  timeout 200 $B 2>/dev/null | line "$tag chunks (cpg)"
  timeout 200 $B --steps 20 --warmup 5 2>/dev/null | line "$tag chunks (cpg, driver args)"
  timeout 200 $B --workload replay 2>/dev/null | line "$tag chunks (replay)"
done
} > gpurun_out/ab.log 2>&1
cat gpurun_out/ab.log
