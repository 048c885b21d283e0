#!/bin/bash
# run one GPU test (pytest node id) against prebuilt variants: gpu_one_test.sh <node id> <variant>...
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T=$1; shift
for lib in "$@"; do
  export NMF_HIP_LIB=$PWD/build/libnmf_$lib.so
  echo "=== $lib"
  timeout 600 python -m pytest "$T" -m gpu -q -x 2>&1 | grep -E "^E |passed|failed" | head -8
done > gpurun_out/one_test.log 2>&1
cat gpurun_out/one_test.log
