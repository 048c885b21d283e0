"""Eye-renderer timing: 4096 flies x 2 eyes, readings only (run through gpurun)."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import numpy as np, torch
import flygym_amd.compose as C
from flygym_amd import HIPSimulation, make_model
from flygym_amd.utils.math import Rotation3D
from flygym_amd.vision import EyeRenderer, Scene

for world_cls, kw in [("FlatGroundWorld", dict(own_body=False)), ("FlatGroundWorld", dict()), ("GappedTerrainWorld", dict()),
                      ("BlocksTerrainWorld", dict()), ("MixedTerrainWorld", dict())]:
    fly, world, _ = make_model()
    if world_cls != "FlatGroundWorld":
        world = getattr(C, world_cls)()
        world.add_fly(fly, (0.4, 0.1, 0.8), Rotation3D("quat", (1, 0, 0, 0)))
    n = 4096
    sim = HIPSimulation(world, n_worlds=n, device=0)
    sim.set_leg_adhesion_states(fly.name, np.ones((n, 6), dtype=np.float32))
    sim.step(300)
    eyes = EyeRenderer(sim, fly.name, Scene(spheres=[(6.0, 4.0, 1.5, 1.0)], sphere_rgb=[(0.9, 0.2, 0.1)], **kw))
    eyes.render(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        eyes.render()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"{world_cls:20s} {kw}: {ms:.2f} ms per {2 * n} eye views = {2 * n * 512 * 450 / ms / 1e9:.3f} T raw-pixel rays/s ({len(eyes.capsule_seg)} body capsules)")
