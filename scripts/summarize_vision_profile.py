"""Condense gpurun_out/prof_<tag>/ (scripts/profile_vision.sh) into profiles/<tag>_*."""
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
out = [f"# rocprofv3 summary `{tag}` — `python scripts/bench_vision.py --steps 200` on 1x MI355X (BASELINE config 3)\n",
       f"bench line under the tracer: {b['value']:.4e} env-steps/s combined; retina kernel {b['roofline']['kernel_ms']:.3f} ms per "
       f"{b['config']['frames_per_tick']} frames = {b['roofline']['achieved']:.0f} GB/s ({100 * b['roofline']['frac']:.1f} % of 8 TB/s)\n"]
stats = find("trace", "kernel_stats.csv")
shutil.copy(stats, dst / f"{tag}_kernel_stats.csv")
out.append("## `--kernel-trace --stats`\n\n| kernel | calls | total ms | avg ms | % |\n|---|---|---|---|---|")
for row in csv.DictReader(open(stats)):
    if float(row["Percentage"]) > 0.05:
        out.append(f"| `{re.sub(r'[(].*', '', row['Name'])[:70]}` | {row['Calls']} | {float(row['TotalDurationNs'])/1e6:.3f} | "
                   f"{float(row['AverageNs'])/1e6:.4f} | {float(row['Percentage']):.2f} |")


def counter(sub, name):
    f = find(sub, "counter_collection.csv")
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "retina_stream" in r["Kernel_Name"] and r["Counter_Name"] == name]
    return sum(vals) / len(vals) if vals else None


fk, wk = counter("pmc_fetch", "FETCH_SIZE"), counter("pmc_write", "WRITE_SIZE")
algo = b["roofline"]["algorithmic_bytes_per_launch"]
out.append(f"\n## HBM traffic of `nmf_retina_stream_kernel` per launch (PMC, separate passes)\n\n"
           f"FETCH_SIZE {fk:.0f} KiB (x2 gfx950 correction for wide coalesced reads -> {2 * fk * 1024 / 1e9:.3f} GB; uncorrected "
           f"{fk * 1024 / 1e9:.3f} GB), WRITE_SIZE {wk:.0f} KiB ({wk * 1024 / 1e9:.3f} GB); algorithmic bytes per launch {algo / 1e9:.3f} GB\n")
(dst / f"{tag}_summary.md").write_text("\n".join(out) + "\n")
(dst / f"{tag}_bench.json").write_text(json.dumps(b) + "\n")
print("\n".join(out))
