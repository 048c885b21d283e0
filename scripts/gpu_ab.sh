#!/bin/bash
# A/B on the GPU box: for each prebuilt variant lib in gpurun_out/variants/*.so run the parity tests + bench
cd $GRAFT_REPO_ROOT
for so in variants/*.so; do
  echo "=== $so"
  cp $so flygym_amd/libnmf_hip.so
  python -m pytest tests -x -q -m gpu 2>&1 | tail -2
  python bench.py --no-cpu-baseline --steps 500 2>&1 | grep '"metric"' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('value %.3e  ms/launch %.2f  iters %.2f contacts %.2f' % (d['value'], d['roofline']['kernel_ms_per_launch'], d['config']['mean_newton_iters'], d['config']['mean_contacts']))"
done
