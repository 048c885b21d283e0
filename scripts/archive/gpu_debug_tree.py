"""Diagnostic (gpurun): general-tree kernel on ALL_BIOLOGICAL vs the oracle, stage by stage."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "oracle"))
import numpy as np, torch
import flygym_amd.compose as C
from flygym_amd import HIPSimulation, anatomy as A
from flygym_amd.utils.math import Rotation3D
import oracle as orc

preset = getattr(A.JointPreset, sys.argv[1] if len(sys.argv) > 1 else "ALL_BIOLOGICAL")
fly = C.Fly(name="t")
sk = A.Skeleton(axis_order=A.AxisOrder.YAW_PITCH_ROLL, joint_preset=preset)
fly.add_joints(sk, neutral_pose=C.KinematicPosePreset.NEUTRAL)
fly.add_actuators(sk.get_actuated_dofs_from_preset("legs_active_only"), C.ActuatorType.POSITION, kp=50.0,
                  neutral_input=C.KinematicPosePreset.NEUTRAL)
fly.add_leg_adhesion()
world = C.FlatGroundWorld()
world.add_fly(fly, (0, 0, 0.8), Rotation3D("quat", (1, 0, 0, 0)))
sim = HIPSimulation(world, n_worlds=2, device=0)
print("nb", sim.model.nb, "nv", sim.model.nv, "star", sim.model["star"])
o = orc.Oracle(sim.model.to_blob(), "f64")
print("reset: seg_xpos err", np.abs(sim.field("seg_xpos").cpu().numpy()[0] - o.arr("seg_xpos")).max(),
      "quat err", np.abs(np.abs(sim.field("seg_xquat").cpu().numpy()[0]) - np.abs(o.arr("seg_xquat"))).max())
sim.set_leg_adhesion_states(fly.name, np.ones((2, 6), dtype=np.float32)); o.ctrl[sim.model.nu - 6:] = 1.0
for k in range(30):
    n = 1 if k < 5 else 20
    sim.step(n); o.step(n)
    q = sim.field("qpos").cpu().numpy()[0]; st = sim.field("stats").cpu().numpy()[0]
    qa = sim.field("qacc").cpu().numpy()[0]
    print(f"t {o.time*1e4:5.0f}: |dq| {np.abs(q - o.qpos).max():.2e} |dqacc| {np.abs(qa - o.arr('qacc')).max():.2e}/{np.abs(o.arr('qacc')).max():.2e} "
          f"ncon {st[0]:.0f}/{o.ints()['ncon']} it {st[1]:.0f}/{o.ints()['solver_iter']} finite {np.isfinite(q).all()}")
torch.cuda.synchronize()
import time
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
s2 = HIPSimulation(world, n_worlds=n, device=0)
s2.set_leg_adhesion_states(fly.name, np.ones((n, 6), dtype=np.float32))
s2.step(300); torch.cuda.synchronize()
t0 = time.time(); s2.step(200); torch.cuda.synchronize(); dt = time.time() - t0
print(f"{n} worlds: {n * 200 / dt:.3e} env-steps/s")
