"""Histogram of Newton iterations / contacts per step over walking worlds (diagnostic, run through gpurun)."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import numpy as np, torch
from flygym_amd import HIPSimulation, make_model
from flygym_amd.controllers import TripodCPG
fly, world, _ = make_model()
n = 4096
sim = HIPSimulation(world, n_worlds=n, device=0)
cpg = TripodCPG(fly.get_actuated_jointdofs_order("position"), 1e-4)
table = cpg.targets(n, 2500, device=sim.device)
ids = sim.replay_ids(fly.name)
sim.set_leg_adhesion_states(fly.name, np.ones((n, 6), dtype=np.float32))
sim.warmup(); sim.step_replay(table, ids, 0, 850)
it, nc = [], []
for k in range(40):
    sim.step_replay(table, ids, 850 + k, 1)
    st = sim.field("stats").cpu().numpy()
    it.append(st[:, 1].copy()); nc.append(st[:, 0].copy())
it, nc = np.concatenate(it).astype(int), np.concatenate(nc).astype(int)
print("iterations histogram", np.bincount(it)[:12], "mean", it.mean())
print("contacts histogram", np.bincount(nc)[:16], "mean", nc.mean())
for c in range(0, 12):
    sel = nc == c
    if sel.any(): print(f" contacts {c}: mean iters {it[sel].mean():.2f} (n {sel.sum()})")
