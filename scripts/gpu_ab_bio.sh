#!/bin/bash
# A/B on the GPU box for libs build/libnmf_<name>.so: ALL_BIOLOGICAL bench lines + its parity tests
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
B="python bench.py --no-cpu-baseline --no-live-counters --joint-preset all_biological"
line() { grep '^{"metric"' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); c = d['config']
    print(sys.argv[1], round(d['value'] / 1e6, 2), 'M', 'ms/launch', round(d['roofline']['kernel_ms_per_launch'], 3), 'iters', round(c['mean_newton_iters'], 3), 'valid', d.get('valid'))
" "$1"; }
{
for lib in "$@"; do
  export NMF_HIP_LIB=$PWD/build/libnmf_$lib.so
  timeout 200 $B 2>/dev/null | line "$lib (all_biological cpg 50)"
  timeout 200 $B --steps 20 --warmup 5 2>/dev/null | line "$lib (all_biological cpg 20)"
  timeout 200 $B --worlds-per-gpu 8192 2>/dev/null | line "$lib (all_biological cpg 50, 8192 worlds)"
done
timeout 600 python -m pytest tests -m gpu -q -k "all_biological or ALL_BIOLOGICAL or general_tree" 2>&1 | tail -4
} > gpurun_out/ab.log 2>&1
cat gpurun_out/ab.log
