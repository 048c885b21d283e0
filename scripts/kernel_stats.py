"""LDS / VGPR / spill figures of every stepping-kernel instantiation in the built library (no GPU needed)."""
import re
import struct
import subprocess
import sys
import tempfile
from pathlib import Path

so = Path(sys.argv[1] if len(sys.argv) > 1 else Path(__file__).resolve().parents[1] / "flygym_amd" / "libnmf_hip.so")
d = so.read_bytes()
i = d.find(b"__CLANG_OFFLOAD_BUNDLE__")
n = struct.unpack_from("<Q", d, i + 24)[0]
off = i + 32
with tempfile.TemporaryDirectory() as tmp:
    for _ in range(n):
        o, s, tl = struct.unpack_from("<QQQ", d, off)
        off += 24
        triple = d[off:off + tl].decode()
        off += tl
        if "gfx950" in triple:
            (Path(tmp) / "co.elf").write_bytes(d[i + o:i + o + s])
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", f"{tmp}/co.elf"], capture_output=True, text=True).stdout
for blk in out.split("- .agpr_count")[1:]:
    name = re.search(r"\.name:\s+(\S+)", blk).group(1)
    if "kernel" not in name:
        continue
    g = lambda k: re.search(r"\." + k + r":\s+(\d+)", blk).group(1)
    demangled = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    print(f"{demangled[:110]:110s} lds {g('group_segment_fixed_size'):>6s} vgpr {g('vgpr_count'):>4s} spill {g('vgpr_spill_count'):>3s} scratch {g('private_segment_fixed_size'):>4s}")
