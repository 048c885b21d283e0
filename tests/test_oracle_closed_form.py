"""Closed-form soft-contact known answers: anchors of the oracle's constraint rows that need no MuJoCo.

A one-body model written out by hand (tests/tiny_models.py: a sphere of mass m over a plane, two coincident end-sphere
contacts, 4 pyramid rows each) has steady states that follow in closed form from MuJoCo's *documented* soft-constraint
model with the reference's parameters (``ContactParams``: solref (2e-4, 1), solimp (0.98, 0.99, 0.5, 0.9999, 2), margin
1e-3, friction 1; reference src/flygym/compose/physics.py:61-111):

    row k of a contact:   f_k = D * max(0, aref_k - J_k a),   aref_k = -B (J_k v) - K d(r) r,   r = distance - margin
    K = 1 / (dmax^2 tc^2 zeta^2),  B = 2 / (dmax tc),  tc = max(solref[0], 2 dt),  d(r) = the solimp curve,
    D = 1 / R,  R = 2 mu^2 (1 - d) / d * (1 + mu^2) * invweight   (pyramidal rows),  J_k = n +- mu t_{1,2}

* at rest (a = v = 0) all 8 rows carry f = -D K d(r) r and 8 f = m g  ->  the penetration r*;
* on a plane tilted by theta, sliding along a pyramid axis, the rows n +- mu t split by the creep velocity v:
  8 D K d |r| = m g cos(theta) and 2 n_c mu^2 D B v = m g sin(theta) while tan(theta) <= mu / 2 (both rows active), the
  uphill row alone up to tan(theta) < mu; no steady state beyond — the stick / slip threshold of the pyramid, mu along
  an axis and mu / (|cos phi| + |sin phi|) in the direction phi between the axes (mu / sqrt(2) on the diagonal).

Every prediction is computed here from those formulas alone (``tiny_models.contact_row_constants``), never from oracle
internals.  The HIP kernel runs the same cases in tests/test_hip_parity_r3.py.
"""

import numpy as np
import pytest
from scipy.optimize import brentq

from tiny_models import contact_row_constants, sphere_on_plane

MASS, RADIUS, MU, G, DT = 1e-3, 0.1, 1.0, 9810.0, 1e-4
SOLREF, SOLIMP, MARGIN = (2e-4, 1.0), (0.98, 0.99, 0.5, 0.9999, 2.0), 1e-3
NC = 2                                                   # coincident end-sphere contacts of the degenerate capsule


def rest_position(normal_load, plain=False, n_rows_active=4):
    """r* with n_c * n_rows * D(r) K d(r) |r| = normal_load."""
    def excess(r):
        Kd, _, D = contact_row_constants(r, MASS, MU, SOLREF, SOLIMP, DT, plain)
        return NC * n_rows_active * D * Kd * (-r) - normal_load
    return brentq(excess, -1e-3, -1e-12, xtol=1e-16)


def plane_frame(n):
    """The engine's contact frame of a plane (oracle / kernel make_frame)."""
    t = np.array([0.0, 1.0, 0.0]) if abs(n[1]) < 0.5 else np.array([0.0, 0.0, 1.0])
    t1 = t - np.dot(t, n) * n
    t1 /= np.linalg.norm(t1)
    return t1, np.cross(n, t1)


def tilted_normal(theta, psi):
    return np.array([np.sin(theta) * np.cos(psi), np.sin(theta) * np.sin(psi), np.cos(theta)])


def downhill_angle(n):
    """Angle phi of the downhill direction in the (t1, t2) plane of the contact frame."""
    t1, t2 = plane_frame(n)
    g_t = np.array([0, 0, -1.0]) - np.dot([0, 0, -1.0], n) * n
    return np.arctan2(np.dot(g_t, t2), np.dot(g_t, t1))


def creep_prediction(theta):
    """(speed, r) of steady sliding along a pyramid axis at slope theta < atan(mu)."""
    N, T = MASS * G * np.cos(theta) / NC, MASS * G * np.sin(theta) / NC       # per contact
    if np.tan(theta) <= MU / 2:                                               # both rows of the sliding axis active
        r = rest_position(MASS * G * np.cos(theta))
        _, B, D = contact_row_constants(r, MASS, MU, SOLREF, SOLIMP, DT)
        return T / (2 * MU * MU * D * B), r
    # the downhill-side row is off: N = D (3 A + B mu v), T = mu D (A + B mu v)  with A = K d |r|
    def excess(r):
        Kd, B, D = contact_row_constants(r, MASS, MU, SOLREF, SOLIMP, DT)
        return Kd * (-r) - (N - T / MU) / (2 * D)
    r = brentq(excess, -1e-3, -1e-12, xtol=1e-16)
    Kd, B, D = contact_row_constants(r, MASS, MU, SOLREF, SOLIMP, DT)
    return (T / (MU * D) - Kd * (-r)) / (B * MU), r


@pytest.mark.parametrize("precision,rtol", [("f64", 1e-6), ("f32", 2e-2)])
@pytest.mark.parametrize("pyramid", ["2mu2", "plain"])
def test_sphere_at_rest_sits_at_the_documented_penetration(oracle_lib, precision, rtol, pyramid):
    from flygym_amd.compiler.model import EngineSemantics

    m = sphere_on_plane(MASS, RADIUS, mu=MU, solref=SOLREF, solimp=SOLIMP, margin=MARGIN, start_height=RADIUS + MARGIN,
                        semantics=EngineSemantics(pyramid_R=pyramid))
    o = oracle_lib.Oracle(m.to_blob(), precision)
    o.step(400)
    vtol = 1e-6 if precision == "f64" else 1e-3         # float32: the position's last bit is 2e-3 of r*
    assert o.ints()["ncon"] == 2 and np.abs(o.qvel).max() < vtol and np.abs(o.arr("qacc")).max() < 1e3 * vtol
    r_star = rest_position(MASS * G, plain=pyramid == "plain")
    r = o.arr("con_dist") - MARGIN
    np.testing.assert_allclose(r, r_star, rtol=rtol)
    f = o.arr("efc_force").reshape(2, 4)
    np.testing.assert_allclose(f, MASS * G / 8, rtol=max(rtol, 1e-7))       # eight equal rows carry the weight
    Kd, _, D = contact_row_constants(r_star, MASS, MU, SOLREF, SOLIMP, DT, pyramid == "plain")
    np.testing.assert_allclose(f, -D * Kd * r_star, rtol=max(rtol, 1e-7))   # f = -D K d(r*) r*
    assert abs(r_star) < 2e-5                                               # the sphere floats inside the margin


@pytest.mark.parametrize("precision,rtol", [("f64", 1e-6), ("f32", 2e-2)])
@pytest.mark.parametrize("slope", [0.3, 0.5, 0.8])
def test_creep_velocity_on_an_incline(oracle_lib, precision, rtol, slope):
    """Soft friction rows let a body on a slope creep at the velocity where the split of the two rows of the sliding axis
    balances gravity (MuJoCo documents this; its noslip post-pass, stripped on the reference's batched path, exists to
    remove it): tan(theta) = 0.3, 0.5 (both rows active, the boundary) and 0.8 (the downhill-side row off)."""
    theta = np.arctan(slope * MU)
    n = tilted_normal(theta, 0.0)                         # tilt about y: t1 = y, downhill along -t2 (a pyramid axis)
    assert abs(abs(downhill_angle(n)) - np.pi / 2) < 1e-12
    m = sphere_on_plane(MASS, RADIUS, normal=n, mu=MU, solref=SOLREF, solimp=SOLIMP, margin=MARGIN, start_height=RADIUS + MARGIN)
    o = oracle_lib.Oracle(m.to_blob(), precision)
    o.step(600)
    v_pred, r_pred = creep_prediction(theta)
    t1, t2 = plane_frame(n)
    v = o.qvel[:3]
    assert np.abs(o.arr("qacc")[:3]).max() < 1e-3 * G                        # steady
    vtol = 1e-6 if precision == "f64" else 2e-2          # float32: the position's last bit is a few 1e-3 of r*
    assert abs(np.dot(v, n)) < vtol * v_pred and abs(np.dot(v, t1)) < vtol * v_pred
    assert -np.dot(v, t2) == pytest.approx(v_pred, rel=rtol)
    np.testing.assert_allclose(o.arr("con_dist") - MARGIN, r_pred, rtol=rtol)
    f = o.arr("efc_force").reshape(2, 4)
    assert ((f[:, 3] > 0) == (slope <= 0.5 - 1e-9)).all() or slope == 0.5     # the downhill-side row switches off beyond mu / 2
    # force balance of the rows: normal = sum, tangential = mu (f2 - f3)
    assert f.sum() == pytest.approx(MASS * G * np.cos(theta), rel=max(rtol, 1e-7))
    assert MU * (f[:, 2] - f[:, 3]).sum() == pytest.approx(MASS * G * np.sin(theta), rel=max(rtol, 1e-7))


@pytest.mark.parametrize("precision,vtol", [("f64", 1e-6), ("f32", 5e-2)])      # float32: the velocity carries the rounding of a position near 0.1
@pytest.mark.parametrize("slope", [0.2, 0.45])
def test_noslip_removes_the_creep(oracle_lib, precision, vtol, slope):
    """The CPU flavour's post-pass (option/noslip_iterations = 5, reference mujoco_globals.yaml:15): friction dimensions
    re-solved without the regulariser, normal forces kept.  The unregularised pair equation is J a = aref = -B v on the
    sliding axis, and a pair of opposing pyramid edges keeps its SUM (at rest: half the normal force), so on a slope with
    tan theta < mu / 2 the creep velocity of the soft rows decays at the rate B (1e4 / s) instead of persisting: the body
    comes to rest at the penetration that carries m g cos(theta), held by a friction force of exactly m g sin(theta).
    (Beyond mu / 2 the pair saturates at its sum and the creep stays: the pyramid's limit, not the pass's.)"""
    theta = np.arctan(slope * MU)
    n = tilted_normal(theta, 0.0)
    m = sphere_on_plane(MASS, RADIUS, normal=n, mu=MU, solref=SOLREF, solimp=SOLIMP, margin=MARGIN, start_height=RADIUS + MARGIN, noslip_iterations=5)
    o = oracle_lib.Oracle(m.to_blob(), precision, cpu_flavour=True)
    o.step(600)
    v_creep, _ = creep_prediction(theta)
    t1, t2 = plane_frame(n)
    assert np.abs(o.qvel[:3]).max() < vtol * v_creep                          # at rest: no creep
    assert np.abs(o.arr("qacc")[:3]).max() < 1e-3 * G
    rtol = 1e-6 if precision == "f64" else 2e-2
    np.testing.assert_allclose(o.arr("con_dist") - MARGIN, rest_position(MASS * G * np.cos(theta)), rtol=rtol)
    f = o.arr("efc_force").reshape(2, 4)
    assert f.sum() == pytest.approx(MASS * G * np.cos(theta), rel=max(rtol, 1e-7))
    assert MU * (f[:, 2] - f[:, 3]).sum() == pytest.approx(MASS * G * np.sin(theta), rel=max(rtol, 1e-6))
    assert (f >= 0).all()


@pytest.mark.parametrize("precision", ["f64", "f32"])
@pytest.mark.parametrize("direction", ["axis", "diagonal", "oblique"])
def test_stick_slip_threshold_of_the_pyramid(oracle_lib, precision, direction):
    """Below tan(theta*) = mu / (|cos phi| + |sin phi|) the body reaches a steady creep; above it no steady state exists and
    it accelerates at about g cos(theta) (tan(theta) - tan(theta*)) or more.  mu = 1: 45 degrees along a pyramid axis,
    35.26 degrees (mu / sqrt(2)) on the diagonal."""
    def normal_for(theta):
        if direction == "axis":
            return tilted_normal(theta, 0.0)
        want = np.pi / 4 if direction == "diagonal" else 3 * np.pi / 8      # |phi| folded into the first quadrant
        fold = lambda psi: np.arctan2(abs(np.sin(downhill_angle(tilted_normal(theta, psi)))),
                                      abs(np.cos(downhill_angle(tilted_normal(theta, psi))))) - want
        hi = min(np.pi / 2 - 0.05, np.arcsin(min(1.0, 0.49 / np.sin(theta))))       # |n_y| < 0.5: the frame's t1 stays y-like
        return tilted_normal(theta, brentq(fold, 0.02, hi))

    phi0 = downhill_angle(normal_for(0.3))
    thr0 = MU / (abs(np.cos(phi0)) + abs(np.sin(phi0)))
    assert thr0 == pytest.approx({"axis": 1.0, "diagonal": 1 / np.sqrt(2), "oblique": 1 / (np.cos(np.pi / 8) + np.sin(np.pi / 8))}[direction], rel=1e-7)
    for factor in (0.9, 1.1):
        theta = np.arctan(factor * thr0)
        n = normal_for(theta)
        phi = downhill_angle(n)
        thr = MU / (abs(np.cos(phi)) + abs(np.sin(phi)))
        assert thr == pytest.approx(thr0, rel=1e-7) and abs(n[1]) < 0.5
        m = sphere_on_plane(MASS, RADIUS, normal=n, mu=MU, solref=SOLREF, solimp=SOLIMP, margin=MARGIN, start_height=RADIUS + MARGIN)
        o = oracle_lib.Oracle(m.to_blob(), precision)
        o.step(1500)
        v1 = np.linalg.norm(o.qvel[:3])
        o.step(1500)
        v2 = np.linalg.norm(o.qvel[:3])
        acc = (v2 - v1) / (1500 * DT)
        if factor < 1:
            assert o.ints()["ncon"] == 2
            assert abs(acc) < 1e-4 * G, f"{direction}: still accelerating below the threshold ({acc:.3g})"
        else:
            floor = G * np.cos(theta) * (np.tan(theta) - thr)
            # (the floor is exact for steady contact along the downhill line; above the threshold the contact chatters — the
            # velocity-dependent rows unload the normal one — and off-axis the pyramid's force is not parallel to the motion,
            # so the body also drifts sideways: 0.6 of the floor separates "accelerates" from "creeps" with a wide margin)
            assert acc > 0.6 * floor, f"{direction}: acceleration {acc:.4g} below the pyramid's floor {floor:.4g}"
            assert acc < G * np.sin(theta)
