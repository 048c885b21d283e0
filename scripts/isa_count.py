"""Static instruction counts of one function of a built library (no GPU needed): a proxy for instruction-count work.
usage: isa_count.py <lib.so> <substring of the demangled function name> [--dump out.s]"""
import re, struct, subprocess, sys, tempfile
from collections import Counter
from pathlib import Path

so, pat = Path(sys.argv[1]), sys.argv[2]
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
    txt = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "-d", "--no-show-raw-insn", "-C", f"{tmp}/co.elf"],
                         capture_output=True, text=True).stdout
blocks = re.split(r"\n(?=[0-9a-f]{16} <)", txt)
for blk in blocks:
    head = blk.split("\n", 1)[0]
    if pat not in head:
        continue
    ops = [ln.split()[0] for ln in blk.split("\n")[1:] if ln.startswith("\t") and ln.split()]
    c = Counter(ops)
    cat = Counter()
    for op, k in c.items():
        key = ("dpp" if "dpp" in op else "valu") if op.startswith("v_") else "lds" if op.startswith("ds_") else \
              "wait" if op in ("s_waitcnt", "s_nop") else "salu" if op.startswith("s_") else "mem"
        cat[key] += k
    print(f"{head[18:130]}\n  total {len(ops)}  " + "  ".join(f"{k} {v}" for k, v in sorted(cat.items())))
    print("  " + "  ".join(f"{op} {k}" for op, k in c.most_common(14)))
    if "--dump" in sys.argv:
        Path(sys.argv[sys.argv.index("--dump") + 1]).write_text(blk)
