#!/bin/bash
# round 6: general actuator + multi-fly tests
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests/test_hip_parity_r6.py -k "general_actuator or two_flies" -m gpu -q -x -s 2>&1 | grep -v Warn | tail -60 ) > gpurun_out/r6c_pytest.log
tail -60 gpurun_out/r6c_pytest.log
