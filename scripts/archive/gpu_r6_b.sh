#!/bin/bash
# round 6: vision tests + the new graph-capture test + sensors
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests/test_hip_parity_r6.py tests/test_sensors.py tests/test_hip_parity_r5.py "tests/test_hip_parity_r3.py::test_config3_vision_at_full_size" -m gpu -q -x -s 2>&1 | grep -v Warn | tail -40 ) > gpurun_out/r6b_pytest.log
tail -40 gpurun_out/r6b_pytest.log
