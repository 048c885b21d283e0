#!/bin/bash
# A/B on the GPU box: for each prebuilt variant lib in variants/*.so run the parity tests, the smoke check (prints the
# deviation from the float64 oracle) and two bench repeats
cd $GRAFT_REPO_ROOT
for so in variants/*.so; do
  echo "=== $so"
  cp $so flygym_amd/libnmf_hip.so
  python -m pytest tests -x -q -m gpu 2>&1 | tail -1
  python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
  for p in legs_only legs_only; do
  python bench.py --no-cpu-baseline --no-live-counters --steps 1000 --joint-preset $p 2>&1 | grep '"metric"' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$p value %.4e  ms/launch %.3f  iters %.2f contacts %.2f' % (d['value'], d['roofline']['kernel_ms_per_launch'], d['config']['mean_newton_iters'], d['config']['mean_contacts']))"
  done
done
