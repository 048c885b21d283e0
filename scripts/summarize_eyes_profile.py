"""Condense gpurun_out/prof_<tag>/ (scripts/profile_eyes.sh: `bench.py --vision render` under rocprofv3) into profiles/<tag>_*."""
import csv, json, re, shutil, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
tag = sys.argv[1]
src, dst = ROOT / "gpurun_out" / f"prof_{tag}", ROOT / "profiles"
find = lambda sub, suffix: (sorted((src / sub).rglob(f"*{suffix}")) or [None])[0]
def bench_line(log):
    for line in (src / log).read_text().splitlines():
        if line.startswith('{"metric"'):
            return json.loads(line)
b = bench_line("bench_trace.log"); r, v = b["roofline"], b["config"]["vision"]
n_views = r["eye_frames_per_launch"]; rays = n_views * r["rays_per_view"]
out = [f"# rocprofv3 summary `{tag}` — `python bench.py --no-cpu-baseline --no-live-counters --no-other-configs --vision render --steps 200` on 1x MI355X (BASELINE config 3, eyes ray-cast)\n",
       f"bench line under the tracer: {b['value']:.4e} env-steps/s combined (physics {v['physics_kernel_ms_per_tick']:.3f} ms + eyes {v['kernel_ms_per_tick']:.3f} ms per "
       f"{v['every_steps']}-step tick); `nmf_eye_kernel`: {r['kernel_ms_per_launch']:.3f} ms per {n_views} eye views x {r['rays_per_view']} rays = "
       f"{r['rays_per_s']:.3e} rays/s; {r['algorithmic_flop_per_ray']} flop per ray -> {r['achieved']:.1f} TFLOP/s = **{100 * r['frac']:.1f} % of the {r['peak']} TFLOP/s f32 vector peak**\n"]
stats = find("trace", "kernel_stats.csv"); shutil.copy(stats, dst / f"{tag}_kernel_stats.csv")
out.append("## `--kernel-trace --stats` (all launches of the process)\n\n| kernel | calls | total ms | avg ms | % |\n|---|---|---|---|---|")
for row in csv.DictReader(open(stats)):
    if float(row["Percentage"]) > 0.05:
        out.append(f"| `{re.sub(r'[(].*', '', row['Name'])[:70]}` | {row['Calls']} | {float(row['TotalDurationNs'])/1e6:.3f} | {float(row['AverageNs'])/1e6:.4f} | {float(row['Percentage']):.2f} |")
trace = find("trace", "kernel_trace.csv")
dur = [(int(x["End_Timestamp"]) - int(x["Start_Timestamp"])) / 1e6 for x in csv.DictReader(open(trace)) if "nmf_eye_kernel" in x["Kernel_Name"]]
out.append(f"\nkernel trace: `nmf_eye_kernel` mean of the last 10 launches {sum(dur[-10:]) / 10:.3f} ms (HIP events of the same run: {r['kernel_ms_per_launch']:.3f} ms)\n")
sq = {}
for sub in ("pmc_sq", "pmc_sq2"):
    f = find(sub, "counter_collection.csv")
    by = {}
    for row in csv.DictReader(open(f)):
        if "nmf_eye_kernel" in row["Kernel_Name"]:
            by.setdefault(row["Dispatch_Id"], {})[row["Counter_Name"]] = float(row["Counter_Value"])
    last = list(by.values())[-10:]
    sq.update({k: sum(d[k] for d in last) / len(last) for k in last[0]})
    shutil.copy(f, dst / f"{tag}_{sub}_counters.csv")
out.append("## SQ counters of `nmf_eye_kernel` (mean of the last 10 launches)\n\n| quantity | value |\n|---|---|")
out.append(f"| VALU instructions per ray (SQ_INSTS_VALU x 64 lanes / rays; issued, all lanes) | {sq['SQ_INSTS_VALU'] * 64 / rays:.1f} |")
out.append(f"| SALU / LDS instructions per ray | {sq['SQ_INSTS_SALU'] * 64 / rays:.1f} / {sq['SQ_INSTS_LDS'] * 64 / rays:.1f} |")
out.append(f"| lanes on per VALU instruction (SQ_THREAD_CYCLES_VALU / (64 SQ_ACTIVE_INST_VALU)) | {sq['SQ_THREAD_CYCLES_VALU'] / (64 * sq['SQ_ACTIVE_INST_VALU']):.3f} |")
out.append(f"| wave cycles with a VALU instruction in flight (SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES) | {sq['SQ_ACTIVE_INST_VALU'] / sq['SQ_WAVE_CYCLES']:.3f} |")
out.append(f"| wave cycles waiting on any counter (SQ_WAIT_ANY / SQ_WAVE_CYCLES) | {sq['SQ_WAIT_ANY'] / sq['SQ_WAVE_CYCLES']:.3f} |")
out.append(f"| wave cycles waiting for an instruction (SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES) | {sq['SQ_WAIT_INST_ANY'] / sq['SQ_WAVE_CYCLES']:.3f} |")
out.append(f"| instruction-cache hit rate | {sq['SQC_ICACHE_HITS'] / sq['SQC_ICACHE_REQ']:.5f} |")
out.append(f"| LDS bank-conflict cycles / LDS active cycles | {sq['SQ_LDS_BANK_CONFLICT'] / max(sq['SQ_LDS_IDX_ACTIVE'], 1):.3f} |")
clock = 2.4e9
valu_rate = sq["SQ_INSTS_VALU"] / (r["kernel_ms_per_launch"] * 1e-3)
out.append(f"| VALU pipe occupancy (instructions/s x 2 cycles per wave64 op / (1024 SIMDs x {clock / 1e9:.1f} GHz)) | {valu_rate * 2 / (1024 * clock):.3f} |")
(dst / f"{tag}_summary.md").write_text("\n".join(out) + "\n")
print("\n".join(out))
