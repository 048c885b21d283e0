#!/bin/bash
# round 3: the whole gpu suite, smoke, and the bench lines of BASELINE configs 2 and 3
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -30 ) > gpurun_out/pytest.log
( timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 ) > gpurun_out/smoke.log
B="python bench.py --no-cpu-baseline --no-live-counters"
{
timeout 300 $B --steps 20 --warmup 5 2>/dev/null | grep '^{"metric"'
timeout 300 $B 2>/dev/null | grep '^{"metric"'
timeout 300 $B --vision resample --steps 200 2>&1 | tail -3
timeout 300 $B --vision render --steps 200 2>&1 | tail -3
} > gpurun_out/bench_lines.log 2>&1
tail -8 gpurun_out/pytest.log; cat gpurun_out/smoke.log; cut -c1-400 gpurun_out/bench_lines.log
