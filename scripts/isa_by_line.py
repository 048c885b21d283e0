"""Static instruction mix of one kernel PER SOURCE LINE (no GPU needed): which lines of the step's straight-line hot path the
scalar-ALU, wait and LDS instructions come from.  Build the library with line tables first:
    scripts/build_variant.sh dbg -DNMF_TOPO_MASK=1 -gline-tables-only
    python scripts/isa_by_line.py build/libnmf_dbg.so "nmf_step_kernel<nmf::HybridTopo<0, 0, 6, 3, 2, 1, 1, 1, 1, 1, 1>, false>" [--top 40] [--kind salu]
Lines are attributed by the innermost inlined frame's file:line as llvm-objdump -l prints them."""
import re, struct, subprocess, sys, tempfile
from collections import Counter, defaultdict
from pathlib import Path

so, pat = Path(sys.argv[1]), sys.argv[2]
top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 40
kind = sys.argv[sys.argv.index("--kind") + 1] if "--kind" in sys.argv else "salu"
d = so.read_bytes()
i = d.find(b"__CLANG_OFFLOAD_BUNDLE__")
n = struct.unpack_from("<Q", d, i + 24)[0]
off = i + 32
with tempfile.TemporaryDirectory() as tmp:
    for _ in range(n):
        o, s, tl = struct.unpack_from("<QQQ", d, off); off += 24
        triple = d[off:off + tl].decode(); off += tl
        if "gfx950" in triple:
            (Path(tmp) / "co.elf").write_bytes(d[i + o:i + o + s])
    txt = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "-d", "-l", "--no-show-raw-insn", "-C", f"{tmp}/co.elf"],
                         capture_output=True, text=True).stdout


def cat(op):
    if op.startswith("v_"): return "dpp" if "dpp" in op else "valu"
    if op.startswith("ds_"): return "lds"
    if op in ("s_waitcnt", "s_nop"): return "wait"
    if op.startswith("s_"): return "salu"
    return "mem"


blocks = re.split(r"\n(?=[0-9a-f]{16} <)", txt)
for blk in blocks:
    head = blk.split("\n", 1)[0]
    if pat not in head:
        continue
    by_line = defaultdict(Counter)
    ops_by_line = defaultdict(Counter)
    cur = "?"
    for ln in blk.split("\n")[1:]:
        m = re.match(r"^; (\S+):(\d+)", ln)
        if m:
            cur = f"{Path(m.group(1)).name}:{m.group(2)}"
            continue
        if ln.startswith("\t") and ln.split():
            op = ln.split()[0]
            by_line[cur][cat(op)] += 1
            ops_by_line[cur][op] += 1
    tot = Counter()
    for c in by_line.values(): tot.update(c)
    print(head[18:120]); print("  total", dict(tot))
    src_cache = {}
    for line, c in sorted(by_line.items(), key=lambda kv: -kv[1][kind])[:top]:
        f, _, no = line.partition(":")
        text = ""
        for base in (Path(__file__).resolve().parents[1] / "flygym_amd" / "csrc",):
            p = base / f
            if p.exists():
                if p not in src_cache: src_cache[p] = p.read_text().splitlines()
                if no.isdigit() and int(no) <= len(src_cache[p]): text = src_cache[p][int(no) - 1].strip()[:110]
        print(f"  {line:22s} {kind} {c[kind]:4d}  (valu {c['valu']:4d} lds {c['lds']:3d} wait {c['wait']:3d})  " + " ".join(f"{o}:{k}" for o, k in ops_by_line[line].most_common(4) if cat(o) == kind) + f"   | {text}")
