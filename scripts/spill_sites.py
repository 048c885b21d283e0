"""Where a kernel of a built library touches scratch (no GPU needed): every scratch_load / scratch_store and call of the
functions whose demangled name contains <pattern>, with the line number inside the function's disassembly.
usage: spill_sites.py <lib.so> <pattern> [--dump out.s]"""
import re, struct, subprocess, sys, tempfile
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
for blk in re.split(r"\n(?=[0-9a-f]{16} <)", txt):
    head = blk.split("\n", 1)[0]
    if pat not in head:
        continue
    lines = blk.split("\n")
    print(head[18:140], "lines", len(lines))
    for k, ln in enumerate(lines):
        if "scratch_" in ln or "s_swappc" in ln:
            print("  ", k, ln.strip().split("//")[0])
    if "--dump" in sys.argv:
        Path(sys.argv[sys.argv.index("--dump") + 1]).write_text(blk)
