#!/bin/bash
# tests + bench lines + the LDS / wait counter pass for the in-tree kernel
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -12 ) > gpurun_out/pytest.log
bash scripts/gpu_ab.sh > /dev/null 2>&1
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_quick; rm -rf $OUT; mkdir -p $OUT
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $OUT -o q -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-live-counters > $OUT/bench.log 2>&1 )
python - <<'P'
import csv, glob, os
f = glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/prof_quick/**/*counter_collection.csv", recursive=True)[0]
by = {}
for r in csv.DictReader(open(f)):
    if "nmf_step_kernel" in r["Kernel_Name"]:
        by.setdefault(r["Dispatch_Id"], {})[r["Counter_Name"]] = float(r["Counter_Value"])
last = list(by.values())[-20:]
acc = {k: sum(d[k] for d in last) / len(last) for k in last[0]}
print({k: f"{v:.4g}" for k, v in acc.items()})
print("bank conflict fraction", acc["SQ_LDS_BANK_CONFLICT"] / acc["SQ_LDS_IDX_ACTIVE"], "valu/step", acc["SQ_INSTS_VALU"] / (4096 * 50), "lds/step", acc["SQ_INSTS_LDS"] / (4096 * 50), "wave cycles/step", 4 * acc["SQ_WAVE_CYCLES"] / (4096 * 50))
P
tail -4 gpurun_out/pytest.log; cat gpurun_out/ab.log
