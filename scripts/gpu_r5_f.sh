#!/bin/bash
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
for v in e7 e6 e8; do echo "== $v"; NMF_HIP_LIB=$GRAFT_REPO_ROOT/build/libnmf_$v.so python scripts/bench_eyes.py 2>&1 | grep "ms per"; done
echo "== tests (e7)"; NMF_HIP_LIB=$GRAFT_REPO_ROOT/build/libnmf_e7.so timeout 900 python -m pytest tests/test_sensors.py -m gpu -q 2>&1 | grep -v Warn | tail -12
} > gpurun_out/r5_eyes_f.txt 2>&1
cat gpurun_out/r5_eyes_f.txt
