#!/bin/bash
cd $GRAFT_REPO_ROOT
for V in "$@"; do echo "== $V"; NMF_HIP_LIB=$PWD/build/libnmf_$V.so python scripts/r4/gpu_qacc_err.py 2>&1 | tail -1; done
