"""Reproduces tests/test_hip_parity_r3.py::test_collapsing_flies... up to a checkpoint, saves the state of one world and prints how the
kernel's step compares with the oracles (debugging aid).  usage: gpu_debug_collapse.py <checkpoint> <world>"""
import sys
import numpy as np
import torch
import flygym_amd.compose as C
from flygym_amd import HIPSimulation, anatomy as A
from flygym_amd.utils.math import Rotation3D
import oracle as orc

cp, w = int(sys.argv[1]), int(sys.argv[2])
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
sim.step(250 * (cp + 1))
keys = ("qpos", "qvel", "ctrl", "qacc_warmstart")
before = {k: sim.field(k)[w].cpu().numpy().astype(np.float64) for k in keys}
np.savez("gpurun_out/collapse_state.npz", **before)
sim.step(1)
torch.cuda.synchronize()
qacc = sim.field("qacc")[w].cpu().numpy().astype(np.float64)
print("kernel stats", sim.field("stats")[w].tolist())
blob = sim.model.to_blob()
open("gpurun_out/collapse_model.blob", "wb").write(blob)
for prec in ("f64", "f32"):
    for mode in ("shared", "documented"):
        r = orc.Oracle(blob, prec)
        r.set_solver_mode(mode)
        for k in keys: r.arr(k)[:] = before[k]
        r.step(1)
        a = r.arr("qacc").astype(np.float64)
        if prec == "f64" and mode == "shared": ref = a.copy()
        print(prec, mode, r.ints(), "max|qacc|", np.abs(a).max(), "dev vs f64 shared", np.abs(a - ref).max())
d = np.abs(qacc - ref)
j = np.argsort(d)[-6:]
print("kernel dev", d.max(), "at dofs", j.tolist(), "vals", qacc[j].tolist(), "ref", ref[j].tolist())
