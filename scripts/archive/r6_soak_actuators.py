import sys; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
from flygym_amd import HIPSimulation
import test_oracle_actuator_types as at
n = 1024
for kind in at.KW:
    fly, world, model = at.fly_with(kind, **at.KW[kind])
    sim = HIPSimulation(world, n_worlds=n, device=0)
    ids = torch.as_tensor([i for i, a in enumerate(fly.actuators) if a["kind"] == kind], device=sim.device)
    g = torch.Generator(device=sim.device); g.manual_seed(1)
    sim.field("qvel")[:, 6:] = (torch.rand((n, sim.model.nv - 6), device=sim.device, generator=g) - 0.5) * 40
    for tick in range(200):
        c = 3.0 * (torch.rand((n, len(ids)), device=sim.device, generator=g) - (0.0 if kind in ("damper", "muscle") else 0.5))
        sim.field("ctrl")[:, ids] = c
        sim.step(100)
    torch.cuda.synchronize()
    ok = bool(torch.isfinite(sim.field("qpos")).all()) and bool(torch.isfinite(sim.field("act")).all())
    print(kind, "steps", n * 20000, "finite", ok, "overflow", sim.overflow_steps(), "z range", float(sim.field("qpos")[:, 2].min()), float(sim.field("qpos")[:, 2].max()),
          "|act| max", float(sim.field("act").abs().max()), "exits", {k: v for k, v in sim.get_solver_exits().items() if v})
