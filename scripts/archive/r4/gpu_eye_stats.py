"""Eye renderer work statistics (needs a library built with -DNMF_EYE_STATS; run through gpurun)."""
import ctypes, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import numpy as np, torch
from flygym_amd import HIPSimulation, make_model, _native
from flygym_amd.vision import EyeRenderer, Scene
fly, world, _ = make_model()
n = 512
sim = HIPSimulation(world, n_worlds=n, device=0)
sim.set_leg_adhesion_states(fly.name, np.ones((n, 6), dtype=np.float32))
sim.step(300)
eyes = EyeRenderer(sim, fly.name, Scene(spheres=[(6.0, 4.0, 1.5, 1.0)], sphere_rgb=[(0.9, 0.2, 0.1)]))
L = _native.lib(); buf = (ctypes.c_ulonglong * 8)()
L.nmf_debug_eye_stats(buf)
eyes.render(); torch.cuda.synchronize()
L.nmf_debug_eye_stats(buf)
g = list(buf)
print(f"groups per view {g[0] / (2 * n):.1f}; group capsule candidates {g[1] / g[0]:.2f}; union of chunk candidates per group {g[2] / g[0]:.2f}; per-chunk candidates {g[3] / (g[0] * 64):.2f}; "
      f"groups that see the sky only {(g[7] & 0xffffffff) / g[0]:.3f}, groups above the horizon without a sphere {(g[7] >> 32) / g[0]:.3f}; "
      f"groups seeing ground {g[4] / g[0]:.2f}, spheres per group {g[5] / g[0]:.2f}, groups with any candidate {g[6] / g[0]:.2f}")
