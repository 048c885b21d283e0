"""Code bytes of every device function in a built library (gfx950 code object): what has to share the instruction cache."""
import re, struct, subprocess, sys, tempfile
from pathlib import Path
so = Path(sys.argv[1]); pat = sys.argv[2] if len(sys.argv) > 2 else ""
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
    txt = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "-s", "-W", "-C", f"{tmp}/co.elf"], capture_output=True, text=True).stdout
rows = []
for ln in txt.splitlines():
    m = re.match(r"\s*\d+:\s+[0-9a-f]+\s+(\d+)\s+FUNC\s+\S+\s+\S+\s+\S+\s+(.*)", ln)
    if m and pat in m.group(2):
        rows.append((int(m.group(1)), m.group(2)))
for sz, name in sorted(rows, reverse=True)[:40]:
    print(f"{sz:8d}  {name[:150]}")
