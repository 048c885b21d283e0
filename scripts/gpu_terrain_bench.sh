#!/bin/bash
# bench lines of the terrain workloads (BASELINE configs 4 / 5 per GPU) and the terrain tests
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_hip_parity.py tests/test_hip_parity_r2.py tests/test_hip_parity_r3.py tests/test_sensors.py -m gpu -q -k "terrain or config4 or config5 or side_faces or sees_the" 2>&1 | tail -6 ) > gpurun_out/pytest_terrain.log
B="python bench.py --no-cpu-baseline --no-live-counters"
line() { grep '^{"metric"' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); c = d['config']
    print(sys.argv[1], round(d['value'] / 1e6, 2), 'M', 'ms/launch', round(d['roofline']['kernel_ms_per_launch'], 3), 'contacts', round(c['mean_contacts'], 2), 'iters', round(c['mean_newton_iters'], 2), 'valid', d.get('valid'))
" "$1"; }
{
for t in gapped blocks mixed; do timeout 300 $B --terrain $t 2>/dev/null | line "$t"; done
timeout 300 $B --terrain mixed --odor --cpg-adhesion 20 2>/dev/null | line "mixed + odor + adhesion (config 5 workload at 4096)"
timeout 300 $B 2>/dev/null | line "flat"
} > gpurun_out/terrain_bench.log 2>&1
cat gpurun_out/pytest_terrain.log gpurun_out/terrain_bench.log
