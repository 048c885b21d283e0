"""HBM load-pattern ceiling for the retina resample (diagnostic, run through gpurun)."""
import ctypes, subprocess, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
lib_path = ROOT / "flygym_amd" / "libretina_mb.so"
if "--build" in sys.argv or not lib_path.exists():
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-x", "hip",
                    str(ROOT / "scripts/retina_microbench.hip"), "-o", str(lib_path)], check=True)
    if "--build" in sys.argv: sys.exit(0)
import torch
L = ctypes.CDLL(str(lib_path))
L.retina_stream.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
n_img, fb = 8192, 512 * 450 * 3
frames = torch.randint(0, 256, (n_img, fb), dtype=torch.uint8, device="cuda")
out = torch.zeros(n_img, dtype=torch.int32, device="cuda")
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for mode, threads, nt in [(0, 512, 1), (0, 512, 0), (0, 256, 1), (1, 512, 1), (1, 512, 0), (1, 256, 1), (1, 1024, 1)]:
    for _ in range(2): assert L.retina_stream(frames.data_ptr(), n_img, fb, out.data_ptr(), mode, threads, nt, st) == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): L.retina_stream(frames.data_ptr(), n_img, fb, out.data_ptr(), mode, threads, nt, st)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(f"mode {'chunk48' if mode == 0 else 'coalesced'} threads {threads} nt {nt}: {ms:.3f} ms  {n_img * fb / ms / 1e9:.2f} TB/s")
