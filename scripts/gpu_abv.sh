#!/bin/bash
# kernel A/B on the GPU box over prebuilt variants build/libnmf_<name>.so (NMF_HIP_LIB): default bench line + the
# driver's arguments for each; the first name also runs the parity tests of the step kernel
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
for lib in "$@"; do
  export NMF_HIP_LIB=$PWD/build/libnmf_$lib.so
  timeout 200 $B 2>/dev/null | line "$lib (cpg)"
  timeout 200 $B --steps 20 --warmup 5 2>/dev/null | line "$lib (cpg, driver args)"
done
export NMF_HIP_LIB=$PWD/build/libnmf_$1.so
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_hip_parity_r2.py -m gpu -q -x 2>&1 | tail -5
} > gpurun_out/abv.log 2>&1
cat gpurun_out/abv.log
