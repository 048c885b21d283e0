#!/bin/bash
# in-tree library on the other skeletons (one bench line each)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for P in legs_active_only all_biological; do
  timeout 300 python bench.py --no-cpu-baseline --no-live-counters --joint-preset $P 2>/dev/null | grep '^{"metric"' | python -c "
import sys, json
d = json.loads(sys.stdin.read()); c = d['config']
print('$P', round(d['value'] / 1e6, 2), 'M', 'ms/launch', round(d['roofline']['kernel_ms_per_launch'], 3), 'contacts', round(c['mean_contacts'], 2), 'iters', round(c['mean_newton_iters'], 2), 'valid', d.get('valid'))"
done > gpurun_out/presets.log 2>&1
cat gpurun_out/presets.log
