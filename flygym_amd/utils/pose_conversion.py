"""Pose bookkeeping helpers (reference ``src/flygym/utils/pose_conversion.py:10-144``): body names, keyframe poses,
``qpos`` -> :class:`~flygym_amd.compose.pose.KinematicPose`.  They take what ``Fly.compile()`` / ``world.compile()``
return here.  The reference's inverse-kinematics fitters (``fit_qpos_to_xpos_xquat``, ``convert_pose_axis_order``:
offline pose tooling on scipy + MuJoCo) are not part of the stepping engine's scope."""

from __future__ import annotations

import numpy as np

__all__ = ["get_body_names", "get_xpos0_xquat0", "qpos_to_kinematic_pose"]


def _full(model):
    return getattr(model, "compiled", model)         # Fly.compile() returns a summary that carries the full model


def get_body_names(model) -> list[str]:
    """Body names in model order; index 0 is the world body, as in MuJoCo."""
    return ["world"] + list(_full(model).meta["seg_names"])


def get_xpos0_xquat0(model, data=None) -> tuple[np.ndarray, np.ndarray]:
    """Positions ``(nbody, 3)`` and orientations ``(nbody, 4)`` (w, x, y, z) of all bodies at the neutral keyframe (the
    world body's row is zero, as MuJoCo reports it before ``mj_forward``)."""
    from ..compiler.rigid import forward_kinematics, quat_mul, quat_normalize

    m = _full(model)
    xpos_b, xmat_b, xquat_b = forward_kinematics(m, np.asarray(m["key_qpos"], dtype=np.float64))
    n = m.nseg
    xpos, xquat = np.zeros((n + 1, 3)), np.zeros((n + 1, 4))
    for s in range(n):
        b = int(m["seg_body"][s])
        xpos[s + 1] = xpos_b[b] + xmat_b[b] @ m["seg_pos"][s]
        xquat[s + 1] = quat_normalize(quat_mul(xquat_b[b], m["seg_quat"][s]))
    return xpos, xquat


def qpos_to_kinematic_pose(model, qpos, axis_order):
    """A pose from a joint-position vector: the left-side and central joints' angles by dof name, the right side filled
    in by mirroring.  ``qpos`` is either the hinge angles alone (a standalone fly) or the engine's full vector with
    the 7 free-joint entries in front."""
    from ..compose.pose import KinematicPose

    m = _full(model)
    names = m.meta["dof_names"]
    q = np.asarray(qpos, dtype=np.float64)
    if q.shape[0] == len(names) + 7:
        q = q[7:]
    if q.shape[0] != len(names):
        raise ValueError(f"expected {len(names)} (or {len(names) + 7}) joint positions, got {q.shape[0]}")
    angles = {}
    for name, value in zip(names, q):
        child = name.split("-")[1]
        if not child.startswith("r"):
            angles[name] = float(value)
    return KinematicPose(joint_angles_rad_dict=angles, axis_order=axis_order, mirror_left2right=True)
