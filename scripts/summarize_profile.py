"""Condense a gpurun_out/prof_<tag>/ directory (scripts/profile_bench.sh) into profiles/<tag>_*.{csv,md}.

The step kernel is launched once per control tick; the bench's timed region is the LAST
`steps/steps_per_launch` launches of nmf_step_kernel in the trace (earlier ones are the reset, the
500-step warm-up and one settle tick), so the summary reports both the all-launch rocprofv3 stats and
the timed-region mean that bench.py's `roofline.kernel_ms_per_launch` must agree with.
"""
import csv, json, re, shutil, sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
tag = sys.argv[1]
src = ROOT / "gpurun_out" / f"prof_{tag}"
dst = ROOT / "profiles"
dst.mkdir(exist_ok=True)


def find(sub, suffix):
    hits = sorted((src / sub).rglob(f"*{suffix}"))
    return hits[0] if hits else None


def bench_line(log):
    for line in (src / log).read_text().splitlines():
        if line.startswith('{"metric"'):
            return json.loads(line)
    return None


out = [f"# rocprofv3 summary `{tag}` — `python bench.py --no-cpu-baseline --no-live-counters` on 1x MI355X\n"]
b = bench_line("bench_trace.log")
n_launch = b["steps"] // b["config"]["steps_per_launch"] * int(b["config"].get("repeats", 1))
out.append(f"bench line under the tracer: value {b['value']:.4e} env-steps/s, kernel_ms_per_launch "
           f"{b['roofline']['kernel_ms_per_launch']:.3f} ms ({b['config']['steps_per_launch']} steps x {b['config']['worlds_per_gpu']} worlds per launch), "
           f"timed region = last {n_launch} launches\n")

stats = find("trace", "kernel_stats.csv")
shutil.copy(stats, dst / f"{tag}_kernel_stats.csv")
out.append("## `--kernel-trace --stats` (all launches of the process)\n\n| kernel | calls | total ms | avg ms | % |\n|---|---|---|---|---|")
for row in csv.DictReader(open(stats)):
    name = re.sub(r"\(.*", "", row["Name"])[:70]
    if float(row["Percentage"]) > 0.01:
        out.append(f"| `{name}` | {row['Calls']} | {float(row['TotalDurationNs'])/1e6:.3f} | {float(row['AverageNs'])/1e6:.4f} | {float(row['Percentage']):.2f} |")

trace = find("trace", "kernel_trace.csv")
rows = [r for r in csv.DictReader(open(trace)) if "nmf_step_kernel" in r["Kernel_Name"]]
dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows]
timed = dur[-n_launch:]
out.append(f"\n## step-kernel launches in dispatch order (ms)\n\n`{[round(d, 3) for d in dur]}`\n")
out.append(f"timed-region mean over the last {n_launch} launches: **{sum(timed)/len(timed):.3f} ms** "
           f"(bench.py HIP-event mean: {b['roofline']['kernel_ms_per_launch']:.3f} ms)\n")
r0 = rows[-1]
out.append("resources as rocprofv3 reports them (its VGPR_Count is not the allocation: the code object's .vgpr_count — "
           "`scripts/kernel_stats.py` — is 256 for this kernel, i.e. two waves per SIMD): "
           f"VGPR {r0.get('VGPR_Count', r0.get('Arch_VGPR_Count','?'))}, accum VGPR {r0.get('Accum_VGPR_Count','?')}, SGPR {r0.get('SGPR_Count','?')}, "
           f"LDS {r0.get('LDS_Block_Size','?')} B/block, scratch {r0.get('Scratch_Size', r0.get('Private_Segment_Size','?'))} B, grid {r0.get('Grid_Size','?')}, workgroup {r0.get('Workgroup_Size','?')}\n")


def counters(sub):
    f = find(sub, "counter_collection.csv")
    if not f:
        return {}
    acc = {}
    rows = [r for r in csv.DictReader(open(f)) if "nmf_step_kernel" in r["Kernel_Name"]]
    by_disp = {}
    for r in rows:
        by_disp.setdefault(r["Dispatch_Id"], {})[r["Counter_Name"]] = float(r["Counter_Value"])
    disp = list(by_disp.values())[-n_launch:]
    for d in disp:
        for k, v in d.items():
            acc[k] = acc.get(k, 0.0) + v / len(disp)
    return acc


fetch, write = counters("pmc_fetch"), counters("pmc_write")
traffic = None
if fetch and write:
    fk, wk = fetch.get("FETCH_SIZE", 0.0), write.get("WRITE_SIZE", 0.0)
    # MI355X_MICROARCH.md §HBM: counters are in KiB; FETCH_SIZE under-reports wide coalesced reads by 2x on
    # gfx950 (this kernel's reads are dword-wide, so the x2 is an upper bound); WRITE_SIZE uncalibrated.
    traffic = (2 * fk + wk) * 1024
    per = b["roofline"].get("algorithmic_bytes_per_env_step", 1928)
    algo = per * b["config"]["worlds_per_gpu"] * b["config"]["steps_per_launch"]
    out.append(f"## HBM traffic per timed launch (PMC, separate passes)\n\nFETCH_SIZE {fk:.1f} KiB (x2 gfx950 correction -> {2*fk*1024/1e6:.2f} MB), "
               f"WRITE_SIZE {wk:.1f} KiB ({wk*1024/1e6:.2f} MB) -> traffic **{traffic/1e6:.2f} MB** per launch; algorithmic bytes "
               f"({per} B x worlds x steps) = {algo/1e6:.2f} MB; compulsory state+table traffic of a {b['config']['steps_per_launch']}-step persistent launch = "
               f"{(per + 168*b['config']['steps_per_launch']) * b['config']['worlds_per_gpu']/1e6:.2f} MB\n")
sq = {**counters("pmc_sq4"), **counters("pmc_sq"), **counters("pmc_sq2"), **counters("pmc_sq3")}
if sq:
    out.append("## SQ counters per timed launch (mean)\n\n| counter | value |\n|---|---|")
    for k in sorted(sq):
        out.append(f"| {k} | {sq[k]:.4g} |")
    if "SQ_WAVE_CYCLES" in sq and sq["SQ_WAVE_CYCLES"]:
        wc = sq["SQ_WAVE_CYCLES"]
        for k in ("SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS"):
            if k in sq:
                out.append(f"| {k} / SQ_WAVE_CYCLES | {sq[k]/wc:.3f} |")
    if sq.get("SQC_ICACHE_REQ"):
        out.append(f"| instruction cache hit rate (SQC_ICACHE_HITS / SQC_ICACHE_REQ) | {sq.get('SQC_ICACHE_HITS', 0.0) / sq['SQC_ICACHE_REQ']:.5f} |")
    if sq.get("SQ_THREAD_CYCLES_VALU") and sq.get("SQ_ACTIVE_INST_VALU"):
        # both in quad-cycle units: thread-cycles / (64 lanes x instruction-cycles) = fraction of lanes a VALU instruction has on
        out.append(f"| active lanes per VALU instruction (SQ_THREAD_CYCLES_VALU / 64 SQ_ACTIVE_INST_VALU) | {sq['SQ_THREAD_CYCLES_VALU'] / (64.0 * sq['SQ_ACTIVE_INST_VALU']):.3f} |")
    if sq.get("SQ_LDS_BANK_CONFLICT") and sq.get("SQ_LDS_IDX_ACTIVE"):
        out.append(f"| SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE | {sq['SQ_LDS_BANK_CONFLICT'] / sq['SQ_LDS_IDX_ACTIVE']:.3f} |")
    out.append("")
(dst / f"{tag}_summary.md").write_text("\n".join(out) + "\n")
(dst / f"{tag}_bench.json").write_text(json.dumps(b, indent=1) + "\n")
if traffic is not None:
    env_steps = b["config"]["worlds_per_gpu"] * b["config"]["steps_per_launch"]
    issue = {}
    if sq.get("SQ_WAVE_CYCLES"):
        # cycles a wave64 VALU instruction holds the SIMD-32 pipe: measured with 4 waves per SIMD issuing independent
        # v_fma_f32 (profiles/valu_issue_microbench.json: 0.60 instructions per cycle per SIMD = 1.66 cycles each)
        mb = ROOT / "profiles" / "valu_issue_microbench.json"
        cyc_per_inst, clock = 2.0, 2.4e9
        if mb.exists():
            t = next(x for x in json.loads(mb.read_text())["tests"] if x["name"].startswith("v_fma_f32 8 independent"))
            cyc_per_inst = 1.0 / max(v["simd_ipc"] for v in t["by_waves_per_simd"].values())
            clock = 1e9 * t["by_waves_per_simd"]["2"]["ghz"]
        issue = {"valu_insts_per_env_step": sq.get("SQ_INSTS_VALU", 0.0) / env_steps,
                 "salu_insts_per_env_step": sq.get("SQ_INSTS_SALU", 0.0) / env_steps,
                 "lds_insts_per_env_step": sq.get("SQ_INSTS_LDS", 0.0) / env_steps,
                 # SQ_* cycle counters tick once per 4 shader cycles (guide: quad-cycles)
                 "wave_cycles_per_env_step": 4.0 * sq["SQ_WAVE_CYCLES"] / env_steps,
                 "valu_active_per_wave": sq.get("SQ_ACTIVE_INST_VALU", 0.0) / sq["SQ_WAVE_CYCLES"],
                 "wait_any_per_wave": sq.get("SQ_WAIT_ANY", 0.0) / sq["SQ_WAVE_CYCLES"],
                 "valu_cycles_per_inst": cyc_per_inst, "shader_clock_hz": clock,
                 "active_lane_fraction": (sq["SQ_THREAD_CYCLES_VALU"] / (64.0 * sq["SQ_ACTIVE_INST_VALU"])
                                          if sq.get("SQ_THREAD_CYCLES_VALU") and sq.get("SQ_ACTIVE_INST_VALU") else None),
                 "lds_bank_conflict_fraction": (sq["SQ_LDS_BANK_CONFLICT"] / sq["SQ_LDS_IDX_ACTIVE"]
                                                if sq.get("SQ_LDS_BANK_CONFLICT") and sq.get("SQ_LDS_IDX_ACTIVE") else None)}
    # bench.py reads hbm_traffic.json for its default (LEGS_ONLY) command; other skeletons keep their own file
    headline = "HybridTopo<0,0," in b["roofline"].get("kernel", "HybridTopo<0,0,")
    rec = {"profile": tag, "traffic_bytes_per_launch": traffic, "worlds_per_gpu": b["config"]["worlds_per_gpu"],
           "steps_per_launch": b["config"]["steps_per_launch"], "control": b["config"].get("control"), "issue": issue,
           "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes, KiB -> bytes, FETCH_SIZE x2 (gfx950), "
                     "mean over the timed-region launches"}
    (dst / f"{tag}_traffic.json").write_text(json.dumps(rec, indent=1) + "\n")
    # bench.py reads hbm_traffic.json for its default (LEGS_ONLY) command.  A persistent launch moves state + outputs
    # once per launch (and once more per chunk hand-off) and one control-table row per step: with a second profile of
    # the same workload at another launch length (argv[2] = its tag) the two terms are fitted per world, so that the
    # bench can scale the figure to any steps_per_launch.
    if headline:
        other = json.loads((dst / f"{sys.argv[2]}_traffic.json").read_text()) if len(sys.argv) > 2 else None
        if other and other["steps_per_launch"] != rec["steps_per_launch"] and other["worlds_per_gpu"] == rec["worlds_per_gpu"]:
            n = rec["worlds_per_gpu"]
            per_step = (rec["traffic_bytes_per_launch"] - other["traffic_bytes_per_launch"]) / (n * (rec["steps_per_launch"] - other["steps_per_launch"]))
            per_launch = rec["traffic_bytes_per_launch"] / n - per_step * rec["steps_per_launch"]
            rec.update(per_world_per_launch_bytes=per_launch, per_world_per_step_bytes=per_step,
                       fitted_from=[tag, other["profile"]], fitted_steps_per_launch=[rec["steps_per_launch"], other["steps_per_launch"]])
        (dst / "hbm_traffic.json").write_text(json.dumps(rec, indent=1) + "\n")
print("\n".join(out))
print("traffic bytes per launch:", traffic)
