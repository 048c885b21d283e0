"""Cycles per articulated-body sweep, in isolation (diagnostic).  usage: aba_microbench.py [--build]"""
import ctypes, subprocess, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
lib_path = ROOT / "flygym_amd" / "libnmf_hip_aba.so"
if "--build" in sys.argv or not lib_path.exists():
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-mllvm", "-amdgpu-sched-strategy=iterative-ilp", "-fPIC", "-shared", f"-I{ROOT/'include'}", f"-I{ROOT/'flygym_amd/csrc'}",
                    "-x", "hip", str(ROOT / "scripts/aba_microbench.hip"), "-o", str(lib_path)], check=True)
    if "--build" in sys.argv: sys.exit(0)
import numpy as np, torch
from flygym_amd import _native
_native.LIB_PATH = lib_path
from flygym_amd import HIPSimulation, make_model
fly, world, _ = make_model()
for n in (1, 2048):
    sim = HIPSimulation(world, n_worlds=n, device=0)
    sim.set_leg_adhesion_states(fly.name, np.ones((n, 6), dtype=np.float32))
    sim.step(600); torch.cuda.synchronize()
    L = _native.lib()
    L.nmf_aba_bench.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    cyc = torch.zeros(n, dtype=torch.int64, device=sim.device)
    for withK in (0, 1):
        L.nmf_aba_bench(sim._batch_h, cyc.data_ptr(), 200, withK)
        c = cyc.cpu().numpy()
        print(f"n_worlds {n:5d} withK {withK}: cycles per ABA  median {np.median(c):.0f}  min {c.min()}  max {c.max()}  (contacts {sim.field('stats')[0,0].item():.0f})")
