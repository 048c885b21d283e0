"""How the contact-space solves end, step by step (GPU): 4096 worlds single-stepped, per step the solve report of every world
(NMF_STATS columns 4..6: report bits, pivots, KKT residual of the last target).  Prints per exit kind the count, the contact /
pivot distribution and the residual quantiles — with the residual test off (option solver=nofallback), so that the residuals of
the ends it would reject are seen.  usage: gpu_exit_hist.py [flat|blocks|mixed] [adhesion] [steps] [preset]"""
import sys
import numpy as np
import torch
import flygym_amd.compose as C
from flygym_amd import HIPSimulation, make_model
from flygym_amd.controllers import TripodCPG
from flygym_amd.utils.math import Rotation3D

terrain = sys.argv[1] if len(sys.argv) > 1 else "flat"
adhesion = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 400
preset = sys.argv[4] if len(sys.argv) > 4 else "legs_only"
fly = make_model(joints_preset=preset)[0]
world = {"flat": C.FlatGroundWorld, "blocks": C.BlocksTerrainWorld, "mixed": C.MixedTerrainWorld}[terrain]()
world.add_fly(fly, (0, 0, 0.8), Rotation3D("quat", (1, 0, 0, 0)))
n = 4096
sim = HIPSimulation(world, n_worlds=n, device=0, _options=dict(solver="nofallback"))
print(sim.batch_info())
cpg = TripodCPG(fly.get_actuated_jointdofs_order("position"), 1e-4)
sim.set_leg_adhesion_states(fly.name, np.ones((n, 6), dtype=np.float32))
if adhesion > 1:
    table = cpg.targets(n, 2500, device="cuda:0", adhesion=(cpg.stance_bins(sim.model, fly), adhesion, 1.0)); ids = sim.replay_ids(fly.name, with_adhesion=True)
else:
    table = cpg.targets(n, 2500, device="cuda:0"); ids = sim.replay_ids(fly.name)
sim.step(500); sim.step_replay(table, ids, 0, 830)
rows = []
for s in range(steps):
    sim.step_replay(table, ids, 830 + s, 1)
    rows.append(sim.field("stats").clone())
st = torch.stack(rows).cpu().numpy().reshape(-1, 8)
bits = st[:, 4].astype(int)
names = ["contact_space", "kkt", "tie", "stall", "cost", "maxiter", "primal", "fallback", "big", "nonoslip", "free"]
print(f"{terrain} adhesion {adhesion} preset {preset}: {len(st)} solves, mean contacts {st[:, 0].mean():.2f}, iterations {st[:, 1].mean():.3f}")
hist = np.bincount(st[:, 0].astype(int), minlength=20)
print("contacts histogram", {i: int(c) for i, c in enumerate(hist) if c})
for k, nm in enumerate(names):
    sel = (bits >> k) & 1 == 1
    if not sel.any():
        continue
    r = st[sel, 6]
    print(f"  {nm:14s} {int(sel.sum()):9d} ({sel.mean() * 100:8.4f} %)  contacts mean {st[sel, 0].mean():5.2f} max {int(st[sel, 0].max()):2d}  pivots mean {st[sel, 5].mean():5.1f} max {int(st[sel, 5].max()):2d}  iters mean {st[sel, 1].mean():.2f} max {int(st[sel, 1].max())}"
          + (f"  resid q50 {np.quantile(r, .5):.1e} q90 {np.quantile(r, .9):.1e} q99 {np.quantile(r, .99):.1e} max {r.max():.1e}  >1e-3: {int((r > 1e-3).sum())}" if k in (2, 3, 4, 5) else ""))
piv = st[(bits & 1) == 1, 5].astype(int)
if piv.size: print("pivots of contact-space solves: q50", int(np.quantile(piv, .5)), "q99", int(np.quantile(piv, .99)), "max", piv.max(), " >40:", int((piv > 40).sum()), ">44:", int((piv > 44).sum()), ">48:", int((piv > 48).sum()), ">52:", int((piv > 52).sum()))
