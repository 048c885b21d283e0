#!/bin/bash
# kernel A/B on the GPU box: bench lines (driver args, default, replay) for the libraries build/libnmf_<name>.so given as args
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
B="python bench.py --no-cpu-baseline --no-live-counters"
line() { grep '^{"metric"' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); c = d['config']
    print(sys.argv[1], round(d['value'] / 1e6, 2), 'M', 'ms/launch', round(d['roofline']['kernel_ms_per_launch'], 3), 'contacts', round(c['mean_contacts'], 2), 'iters', round(c['mean_newton_iters'], 2), 'clock', round((d['roofline']['compute'].get('shader_clock_hz') or 0) / 1e9, 3), 'valid', d.get('valid'))
" "$1"; }
{
for lib in "$@"; do
  export NMF_HIP_LIB=$PWD/build/libnmf_$lib.so
  for rep in 1 2; do
  timeout 200 $B --steps 20 --warmup 5 2>/dev/null | line "$lib (cpg, driver args)"
  done
  timeout 200 $B 2>/dev/null | line "$lib (cpg, 50-step launches)"
  timeout 200 $B --workload replay --steps 20 --warmup 5 2>/dev/null | line "$lib (replay, 20-step launches)"
done
} > gpurun_out/ab.log 2>&1
cat gpurun_out/ab.log
