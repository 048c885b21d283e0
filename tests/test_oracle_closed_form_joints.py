"""Closed-form joint-space anchors of the oracle (round 3): discrete recurrences that follow from MuJoCo's *documented*
actuator, passive-force, integrator and soft-constraint models, computed here without any oracle internals and compared
step for step.

* A hinge on a base 1e9 times heavier (tests/tiny_models.py::hinge_on_heavy_base; no gravity, no contact) is the one-dof
  system the documentation writes down.  Position actuator (reference ``compose/fly.py:351-381``: gain kp, bias
  (0, -kp, -kv), optional forcerange): ``tau = clamp(kp ctrl - kp q - kv qd)``; passive: ``-stiffness (q - springref) -
  damping qd``; Euler with implicit joint damping (``eulerdamp``, the reference's integrator):
  ``qd' = qd + h f / (I + armature + h damping)``, ``q' = q + h qd'``  with f the total force at (q, qd).
* A free body on the tether weld (TetheredWorld: solref (2e-4, 1), solimp (0.98, 0.99, 1e-5, 0.5, 3), reference
  ``compose/world.py:358-365``) under gravity: one bilateral row per axis, whose solution is the documented impedance
  mix ``a = (1 - d) a0 + d aref`` with ``aref = -B v - K d(r) r``, ``a0 = g`` — a recurrence in (r, v) with the solimp curve
  d(r) re-evaluated every step, and a rest offset that solves ``|r| d(r)^2 / (1 - d(r)) = g / K``... (D = m d / (1 - d)).

The HIP kernel runs the same models in tests/test_hip_parity_r3.py.
"""

import numpy as np
import pytest
from scipy.optimize import brentq

from tiny_models import hinge_on_heavy_base, impedance, welded_body

H = 1e-4


def hinge_recurrence(n, q, v, ctrl, inertia, armature, damping, stiffness, springref, kp, kv, forcerange, servo="position"):
    qs, vs, taus = [], [], []
    for k in range(n):
        tau = kp * ctrl(k) - kp * q - kv * v if servo == "position" else kv * (ctrl(k) - v)
        if forcerange is not None:
            tau = min(max(tau, forcerange[0]), forcerange[1])
        f = tau - stiffness * (q - springref) - damping * v
        v = v + H * f / (inertia + armature + H * damping)
        q = q + H * v
        qs.append(q); vs.append(v); taus.append(tau)
    return np.array(qs), np.array(vs), np.array(taus)


HINGE_CASES = {
    # free decay of a spring-damper from 0.3 rad: implicit damping visibly differs from the explicit update here
    # (h damping / I = 0.2)
    "spring-damper decay": dict(par=dict(damping=5e-1, stiffness=2.0, springref=-0.1, armature=1e-6, q0=0.3), ctrl=lambda k: 0.0),
    # position servo step response, clamped by its force range for the first ~2 ms, then tracking a moving target
    "clamped servo": dict(par=dict(kp=50.0, kv=0.0, damping=1e-2, forcerange=(-2.0, 2.0), q0=0.0),
                          ctrl=lambda k: 0.6 + 0.2 * np.sin(2 * np.pi * 12.0 * k * H)),
    # MuJoCo's velocity servo (ActuatorType.VELOCITY: force = kv (ctrl - qd)), clamped at first, against a joint spring
    "velocity servo": dict(par=dict(kv=2e-2, damping=1e-3, stiffness=1.0, springref=0.0, forcerange=(-0.05, 0.05), q0=0.0, servo="velocity"),
                           ctrl=lambda k: 3.0 * np.cos(2 * np.pi * 6.0 * k * H)),
    # servo with velocity feedback and joint spring together
    "servo + spring + kv": dict(par=dict(kp=20.0, kv=5e-3, damping=2e-3, stiffness=5.0, springref=0.2, q0=-0.4), ctrl=lambda k: 0.1),
}


@pytest.mark.parametrize("precision,rtol", [("f64", 1e-7), ("f32", 2e-3)])
@pytest.mark.parametrize("case", list(HINGE_CASES))
def test_hinge_follows_the_documented_recurrence(oracle_lib, case, precision, rtol):
    c = HINGE_CASES[case]
    par = dict(inertia_yy=2e-6, mass=1e-3, com=(0.5, 0.0, 0.0), armature=1e-6, damping=0.0, stiffness=0.0, springref=0.0, kp=0.0, kv=0.0,
               forcerange=None, q0=0.0, servo="position")
    par.update(c["par"])
    m = hinge_on_heavy_base(**par)
    o = oracle_lib.Oracle(m.to_blob(), precision)
    n = 1500
    inertia = par["inertia_yy"] + par["mass"] * float(np.dot(par["com"], par["com"]))
    qs, vs, taus = hinge_recurrence(n, par["q0"], 0.0, c["ctrl"], inertia, par["armature"], par["damping"], par["stiffness"],
                                    par["springref"], par["kp"], par["kv"], par["forcerange"], par["servo"])
    got_q, got_v, got_tau = np.zeros(n), np.zeros(n), np.zeros(n)
    for k in range(n):
        o.ctrl[0] = c["ctrl"](k)
        o.step(1)
        got_q[k], got_v[k], got_tau[k] = o.qpos[7], o.qvel[6], o.arr("actuator_force")[0]
    scale_q, scale_v = np.abs(qs).max(), np.abs(vs).max()
    assert np.abs(got_q - qs).max() < rtol * scale_q, case
    assert np.abs(got_v - vs).max() < rtol * scale_v, case
    assert np.abs(got_tau - taus).max() < max(rtol, 1e-9) * max(np.abs(taus).max(), 1e-12), case
    if par["forcerange"] is not None:
        assert (np.abs(taus[:10]) == par["forcerange"][1]).all() and np.abs(taus).min() < 0.5 * par["forcerange"][1]   # clamp was exercised, and left
    assert np.abs(o.qpos[:3] - [0, 0, 100.0]).max() < 1e-6 and scale_q > 0.05     # the base stayed put, the hinge moved


WELD = dict(solref=(2e-4, 1.0), solimp=(0.98, 0.99, 1e-5, 0.5, 3.0))
G = 9810.0


def weld_constants():
    tc, zeta = max(WELD["solref"][0], 2 * H), WELD["solref"][1]
    dmax = WELD["solimp"][1]
    return 1.0 / (dmax * dmax * tc * tc * zeta * zeta), 2.0 / (dmax * tc)


def weld_recurrence(n, r, v, g):
    K, B = weld_constants()
    rs = []
    for _ in range(n):
        d = impedance(r, WELD["solimp"])
        a = (1.0 - d) * (-g) + d * (-B * v - K * d * r)
        v = v + H * a
        r = r + H * v
        rs.append(r)
    return np.array(rs)


@pytest.mark.parametrize("precision,rtol", [("f64", 1e-6), ("f32", 5e-2)])
def test_welded_body_follows_the_documented_impedance_mix(oracle_lib, precision, rtol):
    """Released 20 um above the weld target under gravity: the z row alone is loaded (the others stay at zero), and the
    trajectory is the recurrence of the documented soft constraint; it ends at the rest offset."""
    off = 2e-5
    m = welded_body(offset=(0.0, 0.0, off), **WELD)
    o = oracle_lib.Oracle(m.to_blob(), precision)
    n = 400
    want = weld_recurrence(n, off, 0.0, G)
    got = np.zeros(n)
    for k in range(n):
        o.step(1)
        got[k] = o.qpos[2] - 100.0
    atol = rtol * off if precision == "f64" else 2e-5       # float32: the position 100 + r carries 7.6e-6 per bit
    assert np.abs(got - want).max() < atol
    K, _ = weld_constants()
    rest = brentq(lambda r: K * impedance(r, WELD["solimp"]) ** 2 / (1.0 - impedance(r, WELD["solimp"])) * (-r) - G, -1e-3, -1e-12, xtol=1e-18)
    assert want[-1] == pytest.approx(rest, rel=1e-3)                      # the recurrence itself has settled on the closed form
    if precision == "f64":
        assert got[-1] == pytest.approx(rest, rel=1e-3) and abs(rest) < 1e-5
        assert np.abs(o.qpos[:2]).max() < 1e-12 and np.abs(o.qpos[4:7]).max() < 1e-12
        f = o.arr("efc_force")
        assert f.shape == (6,) and f[5] == pytest.approx(1e-3 * G, rel=1e-4) and np.abs(f[:5]).max() < 1e-9
