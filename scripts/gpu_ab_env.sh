#!/bin/bash
# A/B on the GPU box: driver-args bench lines for "<lib>[:ENV=VAL,...]" specs; libs are build/libnmf_<lib>.so
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
B="python bench.py --no-cpu-baseline --no-live-counters"
line() { grep '^{"metric"' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); c = d['config']
    print(sys.argv[1], round(d['value'] / 1e6, 2), 'M', 'ms/launch', round(d['roofline']['kernel_ms_per_launch'], 3), 'valid', d.get('valid'))
" "$1"; }
{
for spec in "$@"; do
  lib=${spec%%:*}; envs=""; [ "$spec" != "$lib" ] && envs=$(echo "${spec#*:}" | tr ',' ' ')
  export NMF_HIP_LIB=$PWD/build/libnmf_$lib.so
  env $envs timeout 200 $B --steps 20 --warmup 5 2>/dev/null | line "$spec (cpg 20)"
  env $envs timeout 200 $B --steps 20 --warmup 5 2>/dev/null | line "$spec (cpg 20)"
  env $envs timeout 200 $B 2>/dev/null | line "$spec (cpg 50)"
  env $envs timeout 200 $B --workload replay --steps 20 --warmup 5 2>/dev/null | line "$spec (replay 20)"
done
} > gpurun_out/ab.log 2>&1
cat gpurun_out/ab.log
