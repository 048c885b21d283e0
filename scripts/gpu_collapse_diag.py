#!/usr/bin/env python
"""Diagnostic (GPU box): the scenario of tests/test_hip_parity_r3.py::test_collapsing_flies... over MANY worlds per checkpoint —
how far the kernel's next-step qacc is from the float64 oracle's, next to the float32 oracle's own distance, and where (which dof)."""
import sys
import numpy as np
import torch
sys.path.insert(0, "tests")
import flygym_amd.compose as C
from flygym_amd import HIPSimulation, anatomy as A
from flygym_amd.utils.math import Rotation3D
from oracle import oracle as oracle_lib

n_pick = int(sys.argv[1]) if len(sys.argv) > 1 else 256
fly = C.Fly(name="t")
fly.add_joints(A.Skeleton(axis_order=A.AxisOrder.YAW_PITCH_ROLL, joint_preset=A.JointPreset.ALL_BIOLOGICAL), neutral_pose=C.KinematicPosePreset.NEUTRAL)
world = C.FlatGroundWorld()
world.add_fly(fly, (0, 0, 0.5), Rotation3D("quat", (1, 0, 0, 0)), bodysegs_with_ground_contact="all")
n = 2048
sim = HIPSimulation(world, n_worlds=n, device=0)
g = torch.Generator(device=sim.device); g.manual_seed(3)
q = sim.field("qpos")
q[:, 2] += 0.6 * torch.rand(n, device=sim.device, generator=g)
quat = torch.randn((n, 4), device=sim.device, generator=g)
q[:, 3:7] = quat / quat.norm(dim=1, keepdim=True)
q[:, 2] += 1.0
blob = sim.model.to_blob()
rng = np.random.default_rng(11)
rows = []
for checkpoint in range(4):
    sim.step(250)
    picks = rng.choice(n, size=n_pick, replace=False)
    sel = torch.as_tensor(picks, device=sim.device)
    before = {k: sim.field(k)[sel].cpu().numpy().astype(np.float64) for k in ("qpos", "qvel", "ctrl", "qacc_warmstart")}
    sim.step(1)
    torch.cuda.synchronize()
    qacc, stats, geom = sim.field("qacc").cpu().numpy(), sim.field("stats").cpu().numpy(), sim.field("contact_geom").cpu().numpy()
    for j, w in enumerate(picks):
        ref = {}
        for prec in ("f64", "f32"):
            r = oracle_lib.Oracle(blob, prec)
            for k in ("qpos", "qvel", "ctrl", "qacc_warmstart"): r.arr(k)[:] = before[k][j]
            r.step(1)
            ref[prec] = r
        nc = int(stats[w, 0])
        mine = geom[w, :nc].astype(int).tolist()
        if mine != ref["f64"].ints()["con_geom"] or mine != ref["f32"].ints()["con_geom"]: continue
        a64 = ref["f64"].arr("qacc"); scale = max(np.abs(a64).max(), 1e4)
        d = np.abs(qacc[w] - a64); d32 = np.abs(ref["f32"].arr("qacc") - a64)
        ds = np.abs(ref["f32"].arr("qacc_smooth") - ref["f64"].arr("qacc_smooth")).max() / max(np.abs(ref["f64"].arr("qacc_smooth")).max(), 1e4)
        rows.append((d.max() / scale, d32.max() / scale, checkpoint, int(w), nc, int(stats[w, 1]), int(stats[w, 4]), int(d.argmax()), int(d32.argmax()), float(np.abs(a64).max()), ds,
                     ref["f64"].ints()["solver_iter"], ref["f32"].ints()["solver_iter"]))
rows.sort(reverse=True)
dev = np.array([r[0] for r in rows]); dev32 = np.array([r[1] for r in rows])
print("states compared", len(rows))
for bar in (1e-3, 3e-3, 5e-3, 1e-2, 2e-2):
    print(f"  beyond {bar:.0e} of the scale: kernel {int((dev > bar).sum())}, float32 oracle {int((dev32 > bar).sum())}")
print("  median kernel %.2e, float32 oracle %.2e; kernel/oracle32 ratio median %.2f" % (np.median(dev), np.median(dev32), np.median(dev / np.maximum(dev32, 1e-12))))
print("worst 12 (dev, dev32, checkpoint, world, ncon, iters, report, dof of dev, dof of dev32, max|qacc| f64, f32 qacc_smooth err, oracle iters 64/32):")
for r in rows[:12]: print("  %.2e %.2e cp%d w%d nc%d it%d rep%d dof%d dof%d %.1f %.2e %d/%d" % r)
