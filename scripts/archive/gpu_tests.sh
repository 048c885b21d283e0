#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 1200 python -m pytest tests -m gpu -q "$@" 2>&1 | tail -40 ) > gpurun_out/pytest.log
( timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 ) > gpurun_out/smoke.log
tail -15 gpurun_out/pytest.log; cat gpurun_out/smoke.log
