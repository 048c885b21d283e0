import ctypes, sys, glob
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import numpy as np, torch
from flygym_amd import _native
res = {}
for so in sorted(glob.glob(str(ROOT / "variants/aba_*.so"))):
    import subprocess
    out = subprocess.run([sys.executable, "-c", f"""
import sys, ctypes; sys.path.insert(0, '{ROOT}')
import numpy as np, torch
from flygym_amd import _native
from pathlib import Path
_native.LIB_PATH = Path('{so}')
from flygym_amd import HIPSimulation, make_model
fly, world, _ = make_model()
sim = HIPSimulation(world, n_worlds=1, device=0)
sim.set_leg_adhesion_states(fly.name, np.ones((1, 6), dtype=np.float32))
sim.step(600); torch.cuda.synchronize()
L = _native.lib(); L.nmf_aba_bench.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
cyc = torch.zeros(1, dtype=torch.int64, device=sim.device)
r = []
for k in (0, 1):
    L.nmf_aba_bench(sim._batch_h, cyc.data_ptr(), 200, k); r.append(int(cyc.item()))
print(r)
"""], capture_output=True, text=True)
    print(Path(so).name, out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:])
