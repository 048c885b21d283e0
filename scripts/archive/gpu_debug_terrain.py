"""Ad-hoc GPU diagnostic (run through gpurun): per-step divergence of the HIP engine from the oracles on a terrain world."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "oracle"))
import numpy as np, torch
import flygym_amd.compose as C
from flygym_amd import HIPSimulation
from flygym_amd import anatomy as A
from flygym_amd.controllers import TripodCPG
from flygym_amd.utils.math import Rotation3D
import oracle as orc

cls = getattr(C, sys.argv[1] if len(sys.argv) > 1 else "MixedTerrainWorld")
fly = C.Fly(name="t")
sk = A.Skeleton(axis_order=A.AxisOrder.YAW_PITCH_ROLL, joint_preset=A.JointPreset.LEGS_ONLY)
fly.add_joints(sk, neutral_pose=C.KinematicPosePreset.NEUTRAL)
fly.add_actuators(sk.get_actuated_dofs_from_preset("legs_active_only"), C.ActuatorType.POSITION, kp=50.0,
                  neutral_input=C.KinematicPosePreset.NEUTRAL)
fly.add_leg_adhesion()
world = cls()
world.add_fly(fly, (0.3, 0.2, 0.8), Rotation3D("quat", (1, 0, 0, 0)))
sim = HIPSimulation(world, n_worlds=4, device=0)
o = orc.Oracle(sim.model.to_blob(), "f64"); o32 = orc.Oracle(sim.model.to_blob(), "f32")
order = fly.get_actuated_jointdofs_order(C.ActuatorType.POSITION)
table = TripodCPG(order, 1e-4).targets(1, 2500)
tdev = torch.as_tensor(np.repeat(table, 4, axis=0), device=sim.device)
ids = sim._ids_by_fly[fly.name]["actuators"][C.ActuatorType.POSITION]
sim.set_leg_adhesion_states(fly.name, np.ones((4, 6), dtype=np.float32))
for x in (o, o32): x.ctrl[42:] = 1.0
step = int(sys.argv[2]) if len(sys.argv) > 2 else 10
lo_, hi_ = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (0, 10**9)
def report(tag):
    if not (lo_ <= tag <= hi_): return
    q = sim.field("qpos").cpu().numpy(); st = sim.field("stats").cpu().numpy()[0]
    print(f"{tag:6d} hip-o64 {np.abs(q[0]-o.qpos).max():.2e} hip-o32 {np.abs(q[0]-o32.qpos).max():.2e} o32-o64 {np.abs(o32.qpos-o.qpos).max():.2e}"
          f" dqacc {np.abs(sim.field('qacc').cpu().numpy()[0]-o32.arr('qacc')).max():.2e}/{np.abs(o32.arr('qacc')).max():.2e}"
          f" ncon {st[0]:.0f}/{o.ints()['ncon']}/{o32.ints()['ncon']} it {st[1]:.0f}/{o.ints()['solver_iter']}/{o32.ints()['solver_iter']}")
for k in range(0, 400, step):
    sim.step(step); o.step(step); o32.step(step); report(k + step)
for k in range(0, 300, step):
    sim.step_replay(tdev, ids, k, step)
    for x in (o, o32): x.step_replay(table[0], np.arange(42), k, step)
    report(400 + k + step)
