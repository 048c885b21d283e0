#!/bin/bash
# instruction-cache / issue counters of the step kernel for variant libraries: gpu_r4_icache.sh <variant>...
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for V in "$@"; do
  export NMF_HIP_LIB=$GRAFT_REPO_ROOT/build/libnmf_$V.so
  OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_ic_$V; rm -rf $OUT; mkdir -p $OUT
  ( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_ANY --output-format csv -d $OUT -o q -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-live-counters --steps 20 --warmup 5 $BENCH_EXTRA > $OUT/bench.log 2>&1 )
  python - "$V" <<'P'
import csv, glob, os, sys
f = glob.glob(os.environ["GRAFT_REPO_ROOT"] + f"/gpurun_out/prof_ic_{sys.argv[1]}/**/*counter_collection.csv", recursive=True)[0]
by = {}
for r in csv.DictReader(open(f)):
    if "nmf_step_kernel" in r["Kernel_Name"]:
        by.setdefault(r["Dispatch_Id"], {})[r["Counter_Name"]] = float(r["Counter_Value"])
last = list(by.values())[-20:]
acc = {k: sum(d[k] for d in last) / len(last) for k in last[0]}
n = 4096 * 20
print(sys.argv[1], {k: f"{v:.4g}" for k, v in acc.items()})
print(sys.argv[1], "icache hit", acc["SQC_ICACHE_HITS"] / acc["SQC_ICACHE_REQ"], "miss/step", acc["SQC_ICACHE_MISSES"] / n, "req/step", acc["SQC_ICACHE_REQ"] / n, "valu/step", acc["SQ_INSTS_VALU"] / n, "wave cycles/step", 4 * acc["SQ_WAVE_CYCLES"] / n, "wait_inst/wave_cycles", acc["SQ_WAIT_INST_ANY"] / acc["SQ_WAVE_CYCLES"])
P
done > gpurun_out/r4_icache.log 2>&1
cat gpurun_out/r4_icache.log
