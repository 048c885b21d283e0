"""Regression vectors of the CPU oracle itself (NOT reference outputs: real MuJoCo cannot run here).

They freeze what `oracle/nmf_oracle.c` (float64) computes today for the benchmark model under the tripod-CPG control
table and for the ALL_BIOLOGICAL skeleton settling, so that a later change to the oracle, the model compiler or the
asset pack that alters the physics is caught by `tests/test_oracle_regression.py`.

    python tests/golden/make_oracle_regression.py        # rewrites tests/golden/oracle_regression.npz
"""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "oracle"))
import numpy as np


def trajectories():
    import flygym_amd.compose as C
    import oracle as orc
    from flygym_amd import anatomy as A, make_model
    from flygym_amd.controllers import TripodCPG
    from flygym_amd.utils.math import Rotation3D

    out = {}
    fly, world, _ = make_model()
    m = world.compile_model()
    o = orc.Oracle(m.to_blob(), "f64")
    o.ctrl[42:] = 1.0
    o.step(500)
    out["legs_only_settled_qpos"] = o.qpos.copy()
    table = TripodCPG(fly.get_actuated_jointdofs_order("position"), 1e-4).targets(1, 2500)[0]
    marks = []
    for k in range(5):
        o.step_replay(table, np.arange(42), 200 * k, 200)
        marks.append(np.concatenate([o.qpos, o.qvel]))
    out["legs_only_cpg_state_every_200"] = np.array(marks)
    out["legs_only_sensordata"] = o.arr("sensordata").copy()
    out["legs_only_ncon_iters"] = np.array([o.ints()["ncon"], o.ints()["solver_iter"]])

    fly = C.Fly(name="t")
    fly.add_joints(A.Skeleton(axis_order=A.AxisOrder.YAW_PITCH_ROLL, joint_preset=A.JointPreset.ALL_BIOLOGICAL),
                   neutral_pose=C.KinematicPosePreset.NEUTRAL)
    legs = A.Skeleton(axis_order=A.AxisOrder.YAW_PITCH_ROLL, joint_preset=A.JointPreset.LEGS_ONLY)
    fly.add_actuators(legs.get_actuated_dofs_from_preset("legs_active_only"), C.ActuatorType.POSITION, kp=50.0,
                      neutral_input=C.KinematicPosePreset.NEUTRAL)
    fly.add_leg_adhesion()
    world = C.FlatGroundWorld()
    world.add_fly(fly, (0, 0, 0.8), Rotation3D("quat", (1, 0, 0, 0)))
    o = orc.Oracle(world.compile_model().to_blob(), "f64")
    o.ctrl[42:] = 1.0
    o.step(400)
    out["all_biological_settled_qpos"] = o.qpos.copy()
    return out


if __name__ == "__main__":
    data = trajectories()
    np.savez_compressed(Path(__file__).with_name("oracle_regression.npz"), **data)
    for k, v in data.items():
        print(k, v.shape)
