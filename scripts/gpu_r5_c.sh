#!/bin/bash
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 600 python -m pytest tests -m gpu -q -k "collapsing" 2>&1 | grep -v Warning | tail -12 ) > gpurun_out/r5_pytest_c.log
{ python scripts/gpu_exit_hist.py flat 1 400; python scripts/gpu_exit_hist.py blocks 1 300; python scripts/gpu_exit_hist.py mixed 20 300; python scripts/gpu_exit_hist.py flat 1 200 all_biological; } > gpurun_out/r5_exit_hist.txt 2>&1
cat gpurun_out/r5_pytest_c.log gpurun_out/r5_exit_hist.txt
