"""One-command pinning kit: dump what REAL MuJoCo computes for the benchmark model, so the CPU oracle stops being
"parity unpinned" (DESIGN.md §4).

CANNOT RUN in the build container (no ``mujoco`` / ``dm_control`` wheels, Python 3.10); run it on any box where the
reference's environment is importable (``mujoco==3.6.0`` as pinned in the reference's uv.lock:1335-1336, ``dm_control``,
the reference package ``flygym``) and this repository is on ``PYTHONPATH``::

    python tests/golden/make_mujoco_golden.py            # writes tests/golden/mujoco_golden.npz
    python -m pytest tests/test_oracle_vs_mujoco.py -q   # oracle vs the dump (skipped while the dump is absent)

What it pins (reference call sites: ``mj.mj_step`` src/flygym/simulation.py:74-76, keyframe reset :41,62; the batched
path ``mjw.step`` src/flygym/warp/simulation.py:260-263 runs the same pipeline without noslip, :427-448 — so noslip is
switched off here too):

* compile-time semantics (SURVEY Appendix A "mass H / mode L" rows): per named segment mass, inertial frame, inertia,
  ``body_invweight0``; which bodies ``fusestatic`` kept; fitted capsule sizes; pair margin / solref / solimp / friction
  as compiled; actuator gain / bias / ranges; ``stat.meaninertia``; dof order;
* the dynamics at frozen states: for a list of contact-rich states (the oracle's own frozen regression states from
  ``tests/golden/oracle_regression.npz`` + states along a MuJoCo rollout of the kinematic replay) — inputs ``qpos qvel
  ctrl qacc_warmstart`` and, after ONE ``mj_step``: ``ncon``, the contact list (segment name of the fly geom, distance,
  position, frame), ``nefc``, ``efc_force``, ``qacc``, ``qacc_smooth``, ``qfrc_constraint``, ``actuator_force``,
  ``sensordata``, solver iterations, next ``qpos / qvel``.

The file holds data only (inputs and MuJoCo's outputs), no reference source.
"""

from __future__ import annotations

import argparse
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def build_reference_sim():
    """The reference benchmark model (``src/flygym_demo/benchmark/time_gpu_simulation.py:21-64`` make_model defaults)
    on the reference's CPU ``Simulation`` — rebuilt here with the reference's public API so that Warp is not needed."""
    from flygym import Simulation
    from flygym.anatomy import ActuatedDOFPreset, AxisOrder, JointPreset, Skeleton
    from flygym.compose import ActuatorType, FlatGroundWorld, Fly, KinematicPosePreset
    from flygym.utils.math import Rotation3D

    fly = Fly()
    skeleton = Skeleton(axis_order=AxisOrder.YAW_PITCH_ROLL, joint_preset=JointPreset.LEGS_ONLY)
    fly.add_joints(skeleton, neutral_pose=KinematicPosePreset.NEUTRAL)
    dofs = fly.skeleton.get_actuated_dofs_from_preset(ActuatedDOFPreset.LEGS_ACTIVE_ONLY)
    fly.add_actuators(dofs, actuator_type=ActuatorType.POSITION, kp=50.0, neutral_input=KinematicPosePreset.NEUTRAL)
    fly.add_leg_adhesion()
    world = FlatGroundWorld()
    world.add_fly(fly, (0, 0, 0.8), Rotation3D("quat", (1, 0, 0, 0)))
    sim = Simulation(world)
    sim.mj_model.opt.noslip_iterations = 0          # the batched reference path strips noslip (warp/simulation.py:439-446)
    return fly, world, sim


def short(name: str) -> str:
    return name.split("/")[-1]


def model_dump(mj, sim) -> dict:
    m = sim.mj_model
    out = {}
    names = [short(mj.mj_id2name(m, mj.mjtObj.mjOBJ_BODY, b) or "") for b in range(m.nbody)]
    out["model/body_names"] = np.array(names)
    for k in ("body_mass", "body_ipos", "body_iquat", "body_inertia", "body_invweight0", "body_pos", "body_quat",
              "body_parentid", "body_dofnum", "body_dofadr", "dof_armature", "dof_damping", "dof_invweight0",
              "jnt_stiffness", "qpos_spring", "geom_size", "geom_type", "geom_bodyid", "geom_pos", "geom_quat",
              "geom_rbound", "pair_geom1", "pair_geom2", "pair_margin", "pair_gap", "pair_solref", "pair_solimp",
              "pair_friction", "pair_dim", "actuator_gainprm", "actuator_biasprm", "actuator_ctrlrange",
              "actuator_forcerange", "actuator_trnid", "actuator_trntype", "actuator_gaintype", "actuator_biastype",
              "eq_data", "eq_solref", "eq_solimp", "key_qpos", "key_ctrl", "qpos0"):
        if hasattr(m, k):
            out["model/" + k] = np.array(getattr(m, k))
    out["model/geom_names"] = np.array([short(mj.mj_id2name(m, mj.mjtObj.mjOBJ_GEOM, g) or "") for g in range(m.ngeom)])
    out["model/joint_names"] = np.array([short(mj.mj_id2name(m, mj.mjtObj.mjOBJ_JOINT, j) or "") for j in range(m.njnt)])
    out["model/actuator_names"] = np.array([short(mj.mj_id2name(m, mj.mjtObj.mjOBJ_ACTUATOR, a) or "") for a in range(m.nu)])
    out["model/sizes"] = np.array([m.nq, m.nv, m.nu, m.nbody, m.ngeom, m.njnt, m.npair, m.nsensordata, m.neq])
    out["model/opt"] = np.array([m.opt.timestep, m.opt.tolerance, m.opt.iterations, m.opt.ls_iterations, m.opt.ls_tolerance,
                                 m.opt.noslip_iterations, m.opt.impratio, float(m.opt.cone), float(m.opt.solver),
                                 float(m.opt.integrator), float(m.opt.disableflags), float(m.opt.enableflags)])
    out["model/gravity"] = np.array(m.opt.gravity)
    out["model/meaninertia"] = np.array([m.stat.meaninertia])
    out["mujoco_version"] = np.array([mj.mj_versionString()])
    return out


def step_dump(mj, sim, state) -> dict:
    """One mj_step from (qpos, qvel, ctrl, qacc_warmstart) and everything the oracle can be compared with."""
    m, d = sim.mj_model, sim.mj_data
    qpos, qvel, ctrl, ws = state
    d.qpos[:] = qpos; d.qvel[:] = qvel; d.ctrl[:] = ctrl; d.qacc_warmstart[:] = ws
    d.time = 0.0
    mj.mj_forward(m, d)             # forward first: contacts / forces of THIS state (mj_step = forward + integrate)
    rec = {"qpos": np.array(qpos), "qvel": np.array(qvel), "ctrl": np.array(ctrl), "qacc_warmstart": np.array(ws)}
    rec["ncon"] = np.array([d.ncon])
    rec["nefc"] = np.array([d.nefc])
    seg, dist, pos, frame, efc_adr = [], [], [], [], []
    for c in range(d.ncon):
        con = d.contact[c]
        g = con.geom2 if short(mj.mj_id2name(m, mj.mjtObj.mjOBJ_GEOM, con.geom1) or "") == "ground_plane" else con.geom1
        seg.append(short(mj.mj_id2name(m, mj.mjtObj.mjOBJ_GEOM, g) or ""))
        dist.append(con.dist); pos.append(np.array(con.pos)); frame.append(np.array(con.frame)); efc_adr.append(con.efc_address)
    rec["con_segment"] = np.array(seg)
    rec["con_dist"] = np.array(dist); rec["con_pos"] = np.array(pos).reshape(-1, 3); rec["con_frame"] = np.array(frame).reshape(-1, 9)
    rec["con_efc_address"] = np.array(efc_adr)
    for k in ("efc_force", "efc_aref", "efc_D", "efc_R", "efc_pos", "efc_margin", "qacc", "qacc_smooth", "qfrc_smooth", "qfrc_constraint",
              "qfrc_actuator", "qfrc_passive", "qfrc_bias", "actuator_force", "sensordata", "xpos", "xquat"):
        rec[k] = np.array(getattr(d, k))
    rec["solver_niter"] = np.array(d.solver_niter)
    mj.mj_step(m, d)
    rec["next_qpos"] = np.array(d.qpos); rec["next_qvel"] = np.array(d.qvel)
    return rec


def frozen_states(fly_amd, world_amd):
    """The states the oracle's own regression fixture freezes (tests/golden/oracle_regression.npz), with the controls the
    CPG table held at those marks and a zero warm start."""
    from flygym_amd.controllers import TripodCPG

    gold = np.load(Path(__file__).with_name("oracle_regression.npz"))
    table = TripodCPG(fly_amd.get_actuated_jointdofs_order("position"), 1e-4).targets(1, 2500)[0]
    states = []
    key_ctrl = world_amd.compile_model()["key_ctrl"]
    ctrl0 = np.array(key_ctrl, dtype=np.float64); ctrl0[42:] = 1.0
    nv = 72
    states.append((gold["legs_only_settled_qpos"], np.zeros(nv), ctrl0, np.zeros(nv)))
    for k, row in enumerate(gold["legs_only_cpg_state_every_200"]):
        ctrl = ctrl0.copy(); ctrl[:42] = table[200 * (k + 1) - 1]
        states.append((row[:73], row[73:], ctrl, np.zeros(nv)))
    return states


def rollout_states(mj, sim, fly_amd, n_marks=12, every=75):
    """States along MuJoCo's own rollout of the reference benchmark: adhesion on, 500-step warm-up, kinematic replay."""
    from flygym_amd.replay import ReplayTargetData

    m, d = sim.mj_model, sim.mj_data
    sim.reset()
    table = ReplayTargetData(1e-4, fly_amd.get_actuated_jointdofs_order("position")).make_target_angles_all_worlds(1, 1000)[0]
    d.ctrl[42:] = 1.0
    for _ in range(500):
        mj.mj_step(m, d)
    states = []
    for s in range(n_marks * every):
        d.ctrl[:42] = table[s]
        if s % every == every - 1:
            states.append((np.array(d.qpos), np.array(d.qvel), np.array(d.ctrl), np.array(d.qacc_warmstart)))
        mj.mj_step(m, d)
    return states


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=str(Path(__file__).with_name("mujoco_golden.npz")))
    args = ap.parse_args()
    try:
        import mujoco as mj
    except ImportError as e:
        raise SystemExit(f"needs the reference's environment (mujoco, dm_control, flygym): {e}")
    from flygym_amd import make_model

    fly_amd, world_amd, _ = make_model()
    fly, world, sim = build_reference_sim()
    out = model_dump(mj, sim)
    # the actuator / dof order this repository assumes must be MuJoCo's (state vectors are exchanged verbatim)
    ours = [d.name for d in fly_amd.get_jointdofs_order()]
    theirs = [n for n in out["model/joint_names"].tolist()][1:]
    if ours != theirs:
        raise SystemExit(f"joint order differs from MuJoCo's: first mismatch at {next(i for i, (a, b) in enumerate(zip(ours, theirs)) if a != b)}")
    states = frozen_states(fly_amd, world_amd) + rollout_states(mj, sim, fly_amd)
    for i, st in enumerate(states):
        for k, v in step_dump(mj, sim, st).items():
            out[f"state{i:02d}/{k}"] = v
    out["n_states"] = np.array([len(states)])
    np.savez_compressed(args.out, **out)
    print(f"wrote {args.out}: {len(states)} states, MuJoCo {out['mujoco_version'][0]}")


if __name__ == "__main__":
    main()
