#!/bin/bash
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
NMF_HIP_LIB=$GRAFT_REPO_ROOT/build/libnmf_terr.so timeout 900 python -m pytest tests -m gpu -q -k "terrain or blocks or Blocks or Mixed or mixed or gapped or Gapped or narrower or edge or config4 or config5" 2>&1 | grep -v Warn | tail -12
python scripts/gpu_ab.py --bench="--no-other-configs --terrain blocks" --bench="--no-other-configs --terrain gapped" --bench="--no-other-configs --terrain mixed" --bench="--no-other-configs --terrain mixed --odor --cpg-adhesion 20 --worlds-per-gpu 1024" tree terr
} > gpurun_out/r5_terr.txt 2>&1
cat gpurun_out/r5_terr.txt
