#!/bin/bash
# round-2 GPU call 1: tests, smoke, bench (default + the driver's arguments), issue microbench, stage profile
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 ) > gpurun_out/c1_pytest.log
( timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 ) > gpurun_out/c1_smoke.log
( timeout 300 python bench.py 2>&1 | tail -3 ) > gpurun_out/c1_bench_default.log
( timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -3 ) > gpurun_out/c1_bench_driver.log
( timeout 300 python bench.py --workload replay --no-cpu-baseline 2>&1 | tail -2 ) > gpurun_out/c1_bench_replay.log
( NMF_BENCH_FORCE_DIST=1 timeout 300 python bench.py --no-cpu-baseline --steps 200 2>&1 | tail -2 ) > gpurun_out/c1_bench_dist1.log
( hipcc --offload-arch=gfx950 -O2 -o /tmp/vmb scripts/valu_issue_microbench.hip && timeout 120 /tmp/vmb gpurun_out/valu_issue_microbench.json ) > gpurun_out/c1_microbench.log 2>&1
( timeout 300 python scripts/stage_profile.py --build && timeout 300 python scripts/stage_profile.py 4096 ) > gpurun_out/c1_stage.log 2>&1
tail -5 gpurun_out/c1_pytest.log; cat gpurun_out/c1_smoke.log; cut -c1-600 gpurun_out/c1_bench_driver.log
