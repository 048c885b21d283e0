"""Condense gpurun_out/prof_<tag>/ (scripts/profile_vision_bench.sh: `bench.py --vision resample` under rocprofv3) into
profiles/<tag>_* and profiles/vision_traffic.json (HBM bytes per eye frame of the retina kernel, read by bench.py)."""
import csv, json, re, shutil, sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
tag = sys.argv[1]
src, dst = ROOT / "gpurun_out" / f"prof_{tag}", ROOT / "profiles"


def find(sub, suffix):
    hits = sorted((src / sub).rglob(f"*{suffix}"))
    return hits[0] if hits else None


def bench_line(log):
    for line in (src / log).read_text().splitlines():
        if line.startswith('{"metric"'):
            return json.loads(line)


b = bench_line("bench_trace.log")
r, v = b["roofline"], b["config"]["vision"]
n_frames = r["eye_frames_per_launch"]
out = [f"# rocprofv3 summary `{tag}` — `python bench.py --no-cpu-baseline --no-live-counters --vision resample --steps 200` on 1x MI355X (BASELINE config 3)\n",
       f"bench line under the tracer: {b['value']:.4e} env-steps/s combined (physics {v['physics_kernel_ms_per_tick']:.3f} ms + retina "
       f"{v['kernel_ms_per_tick']:.3f} ms per {v['every_steps']}-step tick); `{r['kernel']}`: {r['kernel_ms_per_launch']:.3f} ms per {n_frames} eye frames "
       f"= {r['achieved']:.0f} GB/s algorithmic = **{100 * r['frac']:.1f} % of 8 TB/s** (HIP events on the launch stream)\n"]
stats = find("trace", "kernel_stats.csv")
shutil.copy(stats, dst / f"{tag}_kernel_stats.csv")
out.append("## `--kernel-trace --stats` (all launches of the process)\n\n| kernel | calls | total ms | avg ms | % |\n|---|---|---|---|---|")
for row in csv.DictReader(open(stats)):
    if float(row["Percentage"]) > 0.05:
        out.append(f"| `{re.sub(r'[(].*', '', row['Name'])[:70]}` | {row['Calls']} | {float(row['TotalDurationNs'])/1e6:.3f} | "
                   f"{float(row['AverageNs'])/1e6:.4f} | {float(row['Percentage']):.2f} |")
trace = find("trace", "kernel_trace.csv")
rows = [x for x in csv.DictReader(open(trace)) if "retina_stream" in x["Kernel_Name"]]
dur = [(int(x["End_Timestamp"]) - int(x["Start_Timestamp"])) / 1e6 for x in rows]
n_timed = b["steps"] // v["every_steps"] * int(b["config"].get("repeats", 1))
timed = dur[-n_timed:]
trace_ms = sum(timed) / len(timed)
out.append(f"\nretina kernel, timed-region mean over the last {n_timed} launches of the trace: **{trace_ms:.4f} ms** "
           f"(bench.py HIP-event mean: {r['kernel_ms_per_launch']:.4f} ms) -> {r['algorithmic_bytes_per_launch'] / trace_ms / 1e6:.0f} GB/s\n")


def counter(sub, name):
    f = find(sub, "counter_collection.csv")
    vals = [float(x["Counter_Value"]) for x in csv.DictReader(open(f)) if "retina_stream" in x["Kernel_Name"] and x["Counter_Name"] == name]
    vals = vals[-n_timed:]
    return sum(vals) / len(vals) if vals else None


fk, wk = counter("pmc_fetch", "FETCH_SIZE"), counter("pmc_write", "WRITE_SIZE")
algo = r["algorithmic_bytes_per_launch"]
traffic = (2 * fk + wk) * 1024
out.append(f"## HBM traffic of `{r['kernel']}` per launch (PMC, separate passes, mean over the timed launches)\n\n"
           f"FETCH_SIZE {fk:.0f} KiB (x2 gfx950 correction for wide coalesced reads, MI355X_MICROARCH.md HBM section -> {2 * fk * 1024 / 1e9:.3f} GB; "
           f"uncorrected {fk * 1024 / 1e9:.3f} GB), WRITE_SIZE {wk:.0f} KiB ({wk * 1024 / 1e9:.3f} GB) -> traffic **{traffic / 1e9:.3f} GB**; "
           f"algorithmic bytes per launch {algo / 1e9:.3f} GB ({n_frames} frames x ({r['algorithmic_bytes_per_eye_frame']['in']} B in + "
           f"{r['algorithmic_bytes_per_eye_frame']['out']} B out)) -> traffic / algorithmic = {traffic / algo:.3f}\n")
(dst / f"{tag}_summary.md").write_text("\n".join(out) + "\n")
(dst / f"{tag}_bench.json").write_text(json.dumps(b, indent=1) + "\n")
(dst / "vision_traffic.json").write_text(json.dumps({
    "profile": tag, "kernel": r["kernel"], "traffic_bytes_per_eye_frame": traffic / n_frames, "traffic_bytes_per_launch": traffic,
    "eye_frames_per_launch": n_frames, "algorithmic_bytes_per_launch": algo, "trace_ms_per_launch": trace_ms,
    "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes, KiB -> bytes, FETCH_SIZE x2 (gfx950), mean over the timed-region launches"}, indent=1) + "\n")
print("\n".join(out))
