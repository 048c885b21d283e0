"""Find the step at which a world of a long walk leaves the physical range (diagnostic, run through gpurun) and save the
state right before it: NMF_SOLVER=<variant> python scripts/r4/gpu_blowup_probe.py <blocks|mixed> [steps]"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import numpy as np, torch
import flygym_amd.compose as C
from flygym_amd import HIPSimulation, make_model
from flygym_amd.controllers import TripodCPG
from flygym_amd.utils.math import Rotation3D
kind = sys.argv[1]; steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
n = 4096
adhesion = 20.0 if kind == "mixed" else 0.0
fly = make_model()[0]
world = {"blocks": C.BlocksTerrainWorld, "mixed": C.MixedTerrainWorld}[kind]()
world.add_fly(fly, (0, 0, 0.8), Rotation3D("quat", (1, 0, 0, 0)))
sim = HIPSimulation(world, n_worlds=n, device=0)
cpg = TripodCPG(fly.get_actuated_jointdofs_order("position"), 1e-4)
table = cpg.targets(n, 2500, device=sim.device, adhesion=(cpg.stance_bins(sim.model, fly), adhesion, 1.0) if adhesion else None)
ids = sim.replay_ids(fly.name, with_adhesion=bool(adhesion))
sim.set_leg_adhesion_states(fly.name, np.ones((n, 6), dtype=np.float32))
sim.warmup()
keys = ("qpos", "qvel", "ctrl", "qacc_warmstart")
bad = lambda: (~torch.isfinite(sim.field("qvel")).all(dim=1)) | (sim.field("qvel").abs().amax(dim=1) > 3e4)
saved = 0
k = 0
while k < steps and saved < 3:
    snap = {key: sim.field(key).clone() for key in keys}
    sim.step_replay(table, ids, k, 50)
    b = bad()
    if bool(b.any()):
        w = int(torch.nonzero(b)[0, 0])
        # replay the 50 steps one at a time from the snapshot (all worlds: the launch is cheap) and catch the step
        for key in keys: sim.field(key)[:] = snap[key]
        for j in range(50):
            before = {key: sim.field(key)[w].cpu().numpy().copy() for key in keys}
            sim.step_replay(table, ids, k + j, 1)
            st = sim.field("stats")[w].cpu().numpy()
            if bool(bad()[w]):
                print(f"world {w} leaves the range at step {k + j}: contacts {st[0]:.0f}, iterations {st[1]:.0f}, overflow {st[2]:.0f}, max |qvel| {float(sim.field('qvel')[w].abs().max()):.3g}, max |qacc| {float(sim.field('qacc')[w].abs().max()):.3g}")
                np.savez(ROOT / "gpurun_out" / f"blowup_{kind}_{saved}.npz", rows=table[w].cpu().numpy(), cur=k + j, world=w, qacc_kernel=sim.field("qacc")[w].cpu().numpy(),
                         geoms=sim.field("contact_geom")[w].cpu().numpy(), **before)
                saved += 1
                break
        else:
            print(f"world {w}: bad after the 50-step launch at {k} but not when replayed step by step (history / schedule dependent)")
        # (the replay re-stepped every world: carry on from here)
        sim.step_replay(table, ids, k + j + 1, 50 - j - 1) if j < 49 else None
        sim.reset_worlds(b.to(torch.uint8)) if hasattr(sim, "reset_worlds") else None
    k += 50
print("done", saved)
