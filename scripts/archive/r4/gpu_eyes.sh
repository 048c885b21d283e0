#!/bin/bash
# eye renderer: parity tests + timing for a variant library
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
V=${1:-eyes}
export NMF_HIP_LIB=$PWD/build/libnmf_$V.so
{
timeout 900 python -m pytest tests/test_sensors.py -m gpu -q -x 2>&1 | grep -E "^E  |passed|failed|Error" | head -10
timeout 300 python bench.py --no-cpu-baseline --no-live-counters --no-other-configs --vision render --steps 200 2>/dev/null | grep '^{"metric"' | python -c "
import sys, json
d = json.loads(sys.stdin.read()); c = d['config']
print('render', round(d['value'] / 1e6, 2), 'M', 'eye kernel ms/tick', round(c['vision']['kernel_ms_per_tick'], 3), 'physics ms/tick', round(c['vision']['physics_kernel_ms_per_tick'], 3), 'rays/s', d['roofline'].get('rays_per_s'))
"
} > gpurun_out/r4_eyes_$V.log 2>&1
cat gpurun_out/r4_eyes_$V.log
