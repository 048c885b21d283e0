#!/bin/bash
cd $GRAFT_REPO_ROOT
rocprofv3 -L 2>/dev/null | grep -o "SQ_THREAD_CYCLES_VALU\|SQ_INST_CYCLES_SALU\|SQ_BUSY_CU_CYCLES\|SQ_INSTS_VMEM\b\|SQ_INSTS_SMEM\|SQ_WAIT_INST_LDS\|SQ_ACTIVE_INST_SCA\|SQ_INST_CYCLES_VALU" | sort | uniq -c > gpurun_out/counters_available.txt
bash scripts/profile_bench.sh r2a_s20 --steps 20 --warmup 5
bash scripts/profile_bench.sh r2a
cat gpurun_out/counters_available.txt
