"""First-principles pins for the CPU oracle's smooth dynamics (the reference holds
no numeric dynamics vectors, so the oracle is anchored on physics identities instead)."""

import numpy as np
import pytest

from flygym_amd.compiler import rigid
from flygym_amd.compiler.model import CompiledModel


def _random_state(o, rng, vel=1.0):
    o.reset()
    o.qpos[7:] += rng.normal(0, 0.4, o.nv - 6)
    q = rng.normal(size=4)
    o.qpos[3:7] = q / np.linalg.norm(q)
    o.qpos[0:3] = rng.normal(0, 1, 3) + np.array([0, 0, 50.0])  # far above the ground
    o.qvel[:] = rng.normal(0, vel, o.nv)


def test_mass_matrix_matches_jacobian_sum(bench_model, oracle_lib):
    _, _, m = bench_model
    o = oracle_lib.Oracle(m.to_blob(), "f64")
    rng = np.random.default_rng(1)
    for _ in range(3):
        _random_state(o, rng)
        o.forward()
        M = np.tril(o.arr("M").reshape(o.nv, o.nv))
        Mref = np.tril(rigid.mass_matrix_from_jacobians(m, o.qpos.copy()))
        np.testing.assert_allclose(M, Mref, rtol=1e-10, atol=1e-16)


def _advance(m, qpos, qvel, qacc, t):
    """State at time t along the constant-generalised-acceleration path (2nd-order exact)."""
    q = qpos.copy()
    v = qvel + qacc * t
    q[0:3] = qpos[0:3] + qvel[0:3] * t + 0.5 * qacc[0:3] * t * t
    rot = qvel[3:6] * t + 0.5 * qacc[3:6] * t * t + np.cross(qvel[3:6], qacc[3:6]) * t ** 3 / 12.0
    ang = np.linalg.norm(rot)
    dq = np.array([1.0, 0, 0, 0]) if ang < 1e-300 else rigid.axis_angle_quat(rot / ang, ang)
    q[3:7] = rigid.quat_mul(qpos[3:7], dq)
    q[7:] = qpos[7:] + qvel[6:] * t + 0.5 * qacc[6:] * t * t
    return q, v


def _body_momenta(m, q, v):
    xpos, xmat, xquat = rigid.forward_kinematics(m, q)
    axis, anchor = rigid.dof_axes_world(m, q, xpos, xmat, xquat)
    out = []
    for b in range(m.nb):
        com = xpos[b] + xmat[b] @ m["body_ipos"][b]
        J = rigid.point_jacobian(m, b, com, axis, anchor)
        Iw = xmat[b] @ rigid.sym6_to_mat(m["body_inertia"][b]) @ xmat[b].T
        out.append((J, J[0:3] @ v, Iw @ (J[3:6] @ v)))
    return out


def test_inverse_dynamics_matches_dalembert(bench_model, oracle_lib):
    """M·q̈ + qfrc_bias equals Σ_b Jvᵀ m (a_c − g) + Jwᵀ d(Iω)/dt with the body motion
    obtained by finite-differencing the kinematics (no shared code with the oracle's RNE)."""
    _, _, m = bench_model
    o = oracle_lib.Oracle(m.to_blob(), "f64")
    rng = np.random.default_rng(2)
    g = m["opt_gravity"]
    for trial in range(2):
        _random_state(o, rng, vel=3.0)
        qacc = rng.normal(0, 50.0, o.nv)
        q0, v0 = o.qpos.copy(), o.qvel.copy()
        o.forward()
        M = o.arr("M").reshape(o.nv, o.nv)
        M = np.tril(M) + np.tril(M, -1).T
        tau_oracle = M @ qacc + o.arr("qfrc_bias")
        eps = 1e-5
        mom_p = _body_momenta(m, *_advance(m, q0, v0, qacc, +eps))
        mom_m = _body_momenta(m, *_advance(m, q0, v0, qacc, -eps))
        mom_0 = _body_momenta(m, q0, v0)
        tau = np.zeros(o.nv)
        for b in range(m.nb):
            J = mom_0[b][0]
            a_c = (mom_p[b][1] - mom_m[b][1]) / (2 * eps)
            Ldot = (mom_p[b][2] - mom_m[b][2]) / (2 * eps)
            tau += J[0:3].T @ (m["body_mass"][b] * (a_c - g)) + J[3:6].T @ Ldot
        tau += m["dof_armature"] * qacc
        scale = np.abs(tau).max()
        np.testing.assert_allclose(tau_oracle, tau, rtol=0, atol=2e-6 * scale)


def test_free_fall_and_momentum(bench_model, oracle_lib):
    """No contact, springs/dampers/actuators off: COM follows a parabola and the angular momentum
    about the COM is conserved, up to the first-order integrator error."""
    _, _, m = bench_model
    m2 = CompiledModel(m)
    for k in ("dof_damping", "dof_stiffness", "act_gain", "act_bias"):
        m2[k] = np.zeros_like(m[k])
    m2["opt_timestep"] = np.array([2e-5])
    o = oracle_lib.Oracle(m2.to_blob(), "f64")
    rng = np.random.default_rng(3)
    _random_state(o, rng, vel=2.0)

    def com_and_momentum():
        mom = _body_momenta(m2, o.qpos.copy(), o.qvel.copy())
        xpos, xmat, _ = rigid.forward_kinematics(m2, o.qpos.copy())
        mass = m2["body_mass"]
        coms = np.array([xpos[b] + xmat[b] @ m2["body_ipos"][b] for b in range(m2.nb)])
        com = (mass[:, None] * coms).sum(0) / mass.sum()
        p = sum(mass[b] * mom[b][1] for b in range(m2.nb))
        L = sum(np.cross(coms[b] - com, mass[b] * mom[b][1]) + mom[b][2] for b in range(m2.nb))
        return com, p, L

    c0, p0, L0 = com_and_momentum()
    n = 500
    o.step(n)
    c1, p1, L1 = com_and_momentum()
    T = n * 2e-5
    mass = m2["body_mass"].sum()
    g = m2["opt_gravity"]
    np.testing.assert_allclose(p1, p0 + mass * g * T, rtol=1e-4, atol=1e-8)  # O(h) integrator error
    # semi-implicit Euler: x_n = x_0 + v_0 T + g T (T + h) / 2
    np.testing.assert_allclose(c1, c0 + p0 / mass * T + 0.5 * g * T * (T + 2e-5), rtol=1e-4, atol=1e-5)
    assert np.linalg.norm(L1 - L0) < 2e-3 * np.linalg.norm(L0)
    assert o.ints()["ncon"] == 0


def test_reference_invariants(bench_model, oracle_lib):
    """The invariants the reference's own tests pin (tests/core/test_simulation.py:59-71,
    98-111, 131-134, 174-180): time = n·dt, qvel = 0 at reset, qpos = neutral at reset,
    unit body quaternions after stepping."""
    fly, _, m = bench_model
    o = oracle_lib.Oracle(m.to_blob(), "f64")
    assert o.time == 0.0
    np.testing.assert_array_equal(o.qvel, 0)
    neutral = np.array([fly.jointdof_to_neutralangle[d] for d in fly.get_jointdofs_order()])
    np.testing.assert_allclose(o.qpos[7:], neutral, atol=1e-12)
    o.step(10)
    assert o.time == pytest.approx(10 * 1e-4, rel=1e-12)
    quats = o.arr("seg_xquat").reshape(-1, 4)
    assert quats.shape[0] == 69
    np.testing.assert_allclose(np.linalg.norm(quats, axis=1), 1.0, atol=1e-9)
