"""Where the headline kernel's scratch accesses sit relative to the elimination's pivot ordinals (no GPU needed): the unrolled
elimination (nmf_dual.h) has one v_rsq_f32 per pivot ordinal, so the n-th v_rsq_f32 after the Gram build marks ordinal n.
usage: spill_map.py <lib.so> [kernel substring]"""
import bisect, re, struct, subprocess, sys, tempfile
from collections import Counter
from pathlib import Path

so = Path(sys.argv[1])
pat = sys.argv[2] if len(sys.argv) > 2 else "nmf_step_kernel<nmf::HybridTopo<0, 0, 6, 3, 2, 1, 1, 1, 1, 1, 1>, false>"
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
    txt = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "-d", "--no-show-raw-insn", "-C", f"{tmp}/co.elf"], capture_output=True, text=True).stdout
for blk in re.split(r"\n(?=[0-9a-f]{16} <)", txt):
    if pat not in blk.split("\n", 1)[0]:
        continue
    lines = blk.split("\n")
    rsq = [k for k, l in enumerate(lines) if "v_rsq_f32" in l]
    gaps = [b - a for a, b in zip(rsq, rsq[1:])]
    # the elimination: the longest run of slowly growing gaps
    start = next((k for k in range(len(gaps) - 8) if all(0 < gaps[k + j + 1] - gaps[k + j] < 12 or abs(gaps[k + j + 1] - gaps[k + j]) < 12 for j in range(8)) and gaps[k] < 200), None)
    where = Counter()
    for k, l in enumerate(lines):
        if "scratch_" not in l:
            continue
        kind = "store" if "store" in l else "load"
        if start is None or k < rsq[start]:
            where[("before the elimination", kind)] += 1
        else:
            o = bisect.bisect(rsq, k) - start - 1
            where[(f"pivot ordinal {o}" if o < len(rsq) - start - 1 else "after the elimination", kind)] += 1
    print(lines[0][18:120], f"\n  {len(lines)} instructions, elimination from line {rsq[start] if start is not None else '?'}, block sizes {gaps[start:start + 4] if start is not None else ''} ...")
    for (w, kind), c in sorted(where.items(), key=lambda x: (x[0][0].startswith("pivot"), int(x[0][0].split()[-1]) if x[0][0].startswith("pivot") else 0, x[0])):
        print(f"  {w:28s} {kind:6s} {c}")
