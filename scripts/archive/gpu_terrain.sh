#!/bin/bash
# round 4: terrain parity tests, the terrain kernels' stage cycles and the terrain / other-skeleton bench lines
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
# optional argument: a variant library build/libnmf_<name>.so instead of the in-tree one
if [ -n "$1" ]; then export NMF_HIP_LIB=$PWD/build/libnmf_$1.so; fi
( timeout 900 python -m pytest tests -m gpu -q -x -k "terrain or probe or edge or hull" 2>&1 | tail -5 )
for t in blocks mixed; do
  python scripts/stage_profile.py 1792 --terrain=$t > gpurun_out/r4_terrain_stage_cycles_$t.txt 2>&1
  head -22 gpurun_out/r4_terrain_stage_cycles_$t.txt | tail -20
done
for extra in "--terrain gapped" "--terrain blocks" "--terrain mixed" "--joint-preset all_biological" "--joint-preset legs_active"; do
  python bench.py --no-cpu-baseline --no-live-counters --no-other-configs $extra 2>/dev/null | grep '^{"metric"' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); c = d['config']
    print('$extra', round(d['value'] / 1e6, 2), 'M', 'contacts', round(c['mean_contacts'], 2), 'iters', round(c['mean_newton_iters'], 2), 'valid', d.get('valid'))
"
done
