#!/usr/bin/env python
"""Kernel A/B on the GPU box (run through gpurun): bench lines — and optionally the smoke check and a test selection — for
library variants, environment switches and sweeps, one parametrised script instead of one shell file per experiment.

    python scripts/gpu_ab.py [options] SPEC [SPEC ...]

SPEC is ``<lib>[:ENV=VAL[,ENV=VAL...]]``; ``<lib>`` names ``build/libnmf_<lib>.so`` (scripts/build_variant.sh) and ``tree``
is the in-tree library.  Every SPEC is run against every ``--bench`` argument string (default: the bench's default
arguments, the driver's ``--steps 20 --warmup 5`` and the replay workload) ``--reps`` times.

    --bench "ARGS"        bench.py arguments of one line (repeatable), e.g. --bench "--terrain blocks --no-other-configs"
    --sweep ENV=a,b,c     cross every SPEC with these values of an environment variable (repeatable: the product is run),
                          e.g. --sweep NMF_CHUNK_DIV=1.4,2,3 --sweep NMF_MAX_CHUNKS=3,4,5,8
    --reps N              repeats of every line (default 1)
    --smoke               run __graft_entry__.smoke() on the first SPEC first (prints the deviation from the oracle)
    --tests "PYTEST ARGS" run this pytest selection (-m gpu) on the first SPEC first, e.g. --tests "tests/test_hip_parity.py -k rollout"
    --counters            also collect the instruction-cache / issue counters of the step kernel (rocprofv3 --pmc, own pass)

Examples (what the round 1-4 one-off scripts did):
    gpu_ab.py tree k2                                       # gpu_ab.sh / gpu_abv.sh / gpu_ab_libs.sh
    gpu_ab.py op:NMF_SOLVER=primal op:NMF_SOLVER=nohist op  # gpu_ab_env.sh / gpu_r4_hist.sh
    gpu_ab.py --bench "--steps 20 --warmup 5" --sweep NMF_CHUNK_DIV=1.4,1.6,2,2.5,3,4 --sweep NMF_MAX_CHUNKS=3,4,5,8 op   # gpu_chunk_sweep4.sh
    gpu_ab.py --smoke --tests "tests/test_hip_parity.py -k 'single_step_parity or rollout_parity'" dual                      # gpu_r4_dual.sh
    gpu_ab.py --counters --bench "--steps 20 --warmup 5 --terrain blocks --no-other-configs" rule tab                       # gpu_r4_icache.sh
"""
import argparse
import csv
import glob
import itertools
import json
import os
import shlex
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(os.environ.get("GRAFT_REPO_ROOT") or Path(__file__).resolve().parents[1])
BENCH = [sys.executable, str(ROOT / "bench.py"), "--no-cpu-baseline", "--no-live-counters"]
DEFAULT_LINES = ["", "--steps 20 --warmup 5", "--workload replay --steps 20 --warmup 5"]
COUNTERS = "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_ANY"


def spec_env(spec, extra):
    lib, _, envs = spec.partition(":")
    env = dict(os.environ)
    if lib != "tree":
        env["NMF_HIP_LIB"] = str(ROOT / "build" / f"libnmf_{lib}.so")
    for kv in filter(None, envs.split(",")):
        k, _, v = kv.partition("=")
        env[k] = v
    env.update(extra)
    if any(k.startswith("NMF_") and k not in ("NMF_HIP_LIB", "NMF_ALLOW_ENV") for k in env):
        env["NMF_ALLOW_ENV"] = "1"      # the library reads its development variables only under this gate
    return env


def bench_line(env, args):
    out = subprocess.run(BENCH + shlex.split(args), env=env, capture_output=True, text=True, timeout=600, cwd=ROOT).stdout
    for ln in out.splitlines():
        if ln.startswith('{"metric"'):
            d = json.loads(ln)
            c, r = d["config"], d["roofline"]
            return (f"{d['value'] / 1e6:7.2f} M  ms/launch {r.get('kernel_ms_per_launch', float('nan')):.3f}  contacts {c.get('mean_contacts', 0):.2f}"
                    f"  iters {c.get('mean_newton_iters', 0):.2f}  clock {((r.get('compute') or {}).get('shader_clock_hz') or 0) / 1e9:.3f} GHz  valid {d.get('valid')}")
    return "no bench line (" + out.strip().splitlines()[-1][:120] + ")" if out.strip() else "no output"


def counters(env, args):
    with tempfile.TemporaryDirectory(dir="/tmp") as tmp:
        cmd = ["rocprofv3", "--kernel-trace", "--pmc", *COUNTERS.split(), "--output-format", "csv", "-d", tmp, "-o", "q", "--", *BENCH, *shlex.split(args)]
        subprocess.run(cmd, env=dict(env, TMPDIR="/tmp"), capture_output=True, text=True, timeout=900, cwd="/tmp")
        files = glob.glob(f"{tmp}/**/*counter_collection.csv", recursive=True)
        if not files:
            return "no counters"
        by = {}
        for r in csv.DictReader(open(files[0])):
            if "nmf_step_kernel" in r["Kernel_Name"]:
                by.setdefault(r["Dispatch_Id"], {})[r["Counter_Name"]] = float(r["Counter_Value"])
    last = list(by.values())[-20:]
    acc = {k: sum(d[k] for d in last) / len(last) for k in last[0]}
    return (f"icache hit {acc['SQC_ICACHE_HITS'] / acc['SQC_ICACHE_REQ']:.4f}  misses/launch {acc['SQC_ICACHE_MISSES']:.3g}  valu/launch {acc['SQ_INSTS_VALU']:.4g}"
            f"  wave cycles/launch {4 * acc['SQ_WAVE_CYCLES']:.4g}  wait_inst/wave_cycles {acc['SQ_WAIT_INST_ANY'] / acc['SQ_WAVE_CYCLES']:.3f}")


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("specs", nargs="+")
    ap.add_argument("--bench", action="append")
    ap.add_argument("--sweep", action="append", default=[])
    ap.add_argument("--reps", type=int, default=1)
    ap.add_argument("--smoke", action="store_true")
    ap.add_argument("--tests")
    ap.add_argument("--counters", action="store_true")
    a = ap.parse_args()
    lines = a.bench or DEFAULT_LINES
    sweeps = [(s.partition("=")[0], s.partition("=")[2].split(",")) for s in a.sweep]
    first = spec_env(a.specs[0], {})
    if a.smoke:
        r = subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.smoke()"], env=first, capture_output=True, text=True, cwd=ROOT, timeout=600)
        print(f"smoke ({a.specs[0]}):", (r.stdout + r.stderr).strip().splitlines()[-1][:200], flush=True)
    if a.tests:
        r = subprocess.run([sys.executable, "-m", "pytest", "-m", "gpu", "-q", "-x", *shlex.split(a.tests)], env=first, capture_output=True, text=True, cwd=ROOT, timeout=1800)
        tail = [ln for ln in r.stdout.splitlines() if ln.startswith("E ") or "passed" in ln or "failed" in ln or "Error" in ln]
        print(f"tests ({a.specs[0]}):", *tail[-12:], sep="\n  ", flush=True)
    for spec in a.specs:
        for combo in itertools.product(*[vals for _, vals in sweeps]):
            extra = {k: v for (k, _), v in zip(sweeps, combo)}
            env = spec_env(spec, extra)
            tag = spec + "".join(f" {k}={v}" for k, v in extra.items())
            for args in lines:
                for _ in range(a.reps):
                    print(f"{tag:32s} [{args or 'default'}] {bench_line(env, args)}", flush=True)
                if a.counters:
                    print(f"{tag:32s} [{args or 'default'}] {counters(env, args)}", flush=True)


if __name__ == "__main__":
    main()
