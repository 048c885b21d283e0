"""Small float64 rigid-body toolkit (numpy) used by the model compiler.

It is deliberately a *different algorithm* from the engine's CRBA/RNE: the mass
matrix here is ``Σ_b J_bᵀ diag(m, I) J_b`` from explicit body Jacobians, so the
tests can use it to cross-check the C oracle and the HIP kernels from first
principles.  Conventions follow MuJoCo's documented ones (quaternions (w,x,y,z);
free joint velocity = world-frame linear + body-frame angular).
"""

from __future__ import annotations

import numpy as np


def quat_normalize(q):
    q = np.asarray(q, dtype=np.float64)
    return q / np.linalg.norm(q)


def quat_mul(a, b):
    aw, ax, ay, az = a
    bw, bx, by, bz = b
    return np.array([
        aw * bw - ax * bx - ay * by - az * bz,
        aw * bx + ax * bw + ay * bz - az * by,
        aw * by - ax * bz + ay * bw + az * bx,
        aw * bz + ax * by - ay * bx + az * bw,
    ])


def quat_to_mat(q):
    w, x, y, z = q
    return np.array([
        [1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
        [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
        [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)],
    ])


def mat_to_quat(m):
    t = np.trace(m)
    if t > 0:
        s = np.sqrt(t + 1.0) * 2
        q = np.array([0.25 * s, (m[2, 1] - m[1, 2]) / s, (m[0, 2] - m[2, 0]) / s, (m[1, 0] - m[0, 1]) / s])
    else:
        i = int(np.argmax(np.diag(m)))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = np.sqrt(m[i, i] - m[j, j] - m[k, k] + 1.0) * 2
        q = np.zeros(4)
        q[0] = (m[k, j] - m[j, k]) / s
        q[1 + i] = 0.25 * s
        q[1 + j] = (m[j, i] + m[i, j]) / s
        q[1 + k] = (m[k, i] + m[i, k]) / s
    if q[0] < 0:
        q = -q
    return q / np.linalg.norm(q)


def axis_angle_quat(axis, angle):
    axis = np.asarray(axis, dtype=np.float64)
    s = np.sin(0.5 * angle)
    return np.array([np.cos(0.5 * angle), *(axis * s)])


def sym6_to_mat(s):
    """(xx, yy, zz, xy, xz, yz) → 3x3."""
    return np.array([[s[0], s[3], s[4]], [s[3], s[1], s[5]], [s[4], s[5], s[2]]])


def mat_to_sym6(m):
    return np.array([m[0, 0], m[1, 1], m[2, 2], m[0, 1], m[0, 2], m[1, 2]])


def forward_kinematics(model, qpos):
    """World poses of the dynamic bodies: (xpos[nb,3], xmat[nb,3,3], xquat[nb,4])."""
    nb = model["body_parent"].shape[0]
    xpos = np.zeros((nb, 3))
    xquat = np.zeros((nb, 4))
    xmat = np.zeros((nb, 3, 3))
    for b in range(nb):
        p = int(model["body_parent"][b])
        if p < 0:
            pos = np.array(qpos[0:3], dtype=np.float64)
            quat = quat_normalize(qpos[3:7])
        else:
            pos = xpos[p] + xmat[p] @ model["body_pos"][b]
            quat = quat_mul(xquat[p], model["body_quat"][b])
            adr, num = int(model["body_dofadr"][b]), int(model["body_dofnum"][b])
            for d in range(adr, adr + num):
                quat = quat_mul(quat, axis_angle_quat(model["dof_axis"][d], qpos[d + 1]))
        quat = quat_normalize(quat)
        xpos[b], xquat[b], xmat[b] = pos, quat, quat_to_mat(quat)
    return xpos, xmat, xquat


def dof_axes_world(model, qpos, xpos, xmat, xquat):
    """World-frame hinge axes and anchors for every dof (root rows: body axes)."""
    nv = model["dof_body"].shape[0]
    axis = np.zeros((nv, 3))
    anchor = np.zeros((nv, 3))
    for b in range(model["body_parent"].shape[0]):
        adr, num = int(model["body_dofadr"][b]), int(model["body_dofnum"][b])
        p = int(model["body_parent"][b])
        if p < 0:
            for i in range(3):
                axis[i] = np.eye(3)[i]
                axis[3 + i] = xmat[b][:, i]
            anchor[0:6] = xpos[b]
            continue
        quat = quat_mul(xquat[p], model["body_quat"][b])
        for d in range(adr, adr + num):
            # axis is fixed in the frame reached after the previous hinges of this body
            axis[d] = quat_to_mat(quat_normalize(quat)) @ model["dof_axis"][d]
            anchor[d] = xpos[b]
            quat = quat_mul(quat, axis_angle_quat(model["dof_axis"][d], qpos[d + 1]))
    return axis, anchor


def point_jacobian(model, body, point, axis, anchor):
    """6 x nv Jacobian (translational rows 0:3 at ``point``, rotational rows 3:6)."""
    nv = model["dof_body"].shape[0]
    J = np.zeros((6, nv))
    b = body
    while b >= 0:
        adr, num = int(model["body_dofadr"][b]), int(model["body_dofnum"][b])
        if int(model["body_parent"][b]) < 0:
            J[0:3, 0:3] = np.eye(3)
            for i in range(3):
                J[3:6, 3 + i] = axis[3 + i]
                J[0:3, 3 + i] = np.cross(axis[3 + i], point - anchor[3 + i])
        else:
            for d in range(adr, adr + num):
                J[3:6, d] = axis[d]
                J[0:3, d] = np.cross(axis[d], point - anchor[d])
        b = int(model["body_parent"][b])
    return J


def mass_matrix_from_jacobians(model, qpos):
    """M(q) = Σ_b Jvᵀ m Jv + Jwᵀ I_world Jw  (+ armature on the diagonal)."""
    xpos, xmat, xquat = forward_kinematics(model, qpos)
    axis, anchor = dof_axes_world(model, qpos, xpos, xmat, xquat)
    nv = model["dof_body"].shape[0]
    M = np.zeros((nv, nv))
    for b in range(model["body_parent"].shape[0]):
        com = xpos[b] + xmat[b] @ model["body_ipos"][b]
        J = point_jacobian(model, b, com, axis, anchor)
        Iw = xmat[b] @ sym6_to_mat(model["body_inertia"][b]) @ xmat[b].T
        M += model["body_mass"][b] * J[0:3].T @ J[0:3] + J[3:6].T @ Iw @ J[3:6]
    M[np.diag_indices(nv)] += model["dof_armature"]
    return M
