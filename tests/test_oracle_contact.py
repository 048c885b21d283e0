"""Known-answer tests for the oracle's contact pipeline (collision, constraint rows, Newton solve,
adhesion, sensors)."""

import numpy as np
import pytest


@pytest.fixture(scope="module")
def settled(bench_model, oracle_lib):
    _, _, m = bench_model
    o = oracle_lib.Oracle(m.to_blob(), "f64")
    o.ctrl[42:] = 1.0
    o.step(1500)
    return m, o


def test_static_force_balance(settled):
    """At rest the ground reaction carries the weight plus the six adhesion pulls (gain 1, ctrl 1)."""
    m, o = settled
    sd = o.arr("sensordata").reshape(6, 16)
    weight = m["body_mass"].sum() * 9810.0
    assert np.abs(o.qvel).max() < 1.0
    assert sd[:, 3].sum() == pytest.approx(weight + 6.0, rel=2e-3)
    assert (sd[:, 0] >= 1).all()                       # all six legs in contact
    # left/right symmetry of the model and of the neutral pose
    np.testing.assert_allclose(sd[:3, 3], sd[3:, 3], rtol=5e-3)
    np.testing.assert_allclose(sd[:3, 2], -sd[3:, 2], atol=5e-3)
    # normals / tangents of the flat ground frame
    np.testing.assert_allclose(sd[:, 10:13], np.tile([0, 0, 1.0], (6, 1)))
    np.testing.assert_allclose(sd[:, 13:16], np.tile([0, 1.0, 0], (6, 1)))


def test_adhesion_off_still_pulls_with_gain_times_one(bench_model, oracle_lib):
    """ctrlrange (1,100) clamps ctrl=0 to 1 (reference fly.py:434-440)."""
    _, _, m = bench_model
    o = oracle_lib.Oracle(m.to_blob(), "f64")
    o.ctrl[42:] = 0.0
    o.step(1500)
    sd = o.arr("sensordata").reshape(6, 16)
    assert sd[:, 3].sum() == pytest.approx(m["body_mass"].sum() * 9810.0 + 6.0, rel=2e-3)
    np.testing.assert_allclose(o.arr("actuator_force")[42:], 1.0)


def test_contact_forces_are_consistent(settled):
    m, o = settled
    i = o.ints()
    f = o.arr("efc_force")
    assert i["nefc"] == 4 * i["ncon"] and (f >= 0).all()
    J = o.arr("J").reshape(i["nefc"], o.nv)
    np.testing.assert_allclose(J.T @ f, o.arr("qfrc_constraint"), rtol=1e-9, atol=1e-12)
    # optimality: M(a - a0) = J^T f at the solver's fixed point
    M = o.arr("M").reshape(o.nv, o.nv)
    M = np.tril(M) + np.tril(M, -1).T
    res = M @ (o.arr("qacc") - o.arr("qacc_smooth")) - o.arr("qfrc_constraint")
    assert np.abs(res).max() < 1e-6 * max(1.0, np.abs(o.arr("qfrc_constraint")).max())
    # friction pyramid: |tangential| <= mu * normal for every contact
    fr = f.reshape(-1, 4)
    fn = fr.sum(1)
    ft = np.hypot(fr[:, 0] - fr[:, 1], fr[:, 2] - fr[:, 3])
    assert (ft <= fn * np.sqrt(2) + 1e-12).all()


def test_f32_oracle_tracks_f64(bench_model, oracle_lib):
    _, _, m = bench_model
    a = oracle_lib.Oracle(m.to_blob(), "f64")
    b = oracle_lib.Oracle(m.to_blob(), "f32")
    for o in (a, b):
        o.ctrl[42:] = 1.0
        o.step(300)
    assert np.abs(a.qpos - b.qpos).max() < 1e-4
    assert a.ints()["ncon"] == b.ints()["ncon"]


def test_hull_manifold_gives_up_to_four_contacts(bench_model, oracle_lib):
    """A fly dropped flat on its belly: thorax/abdomen hulls produce multi-point manifolds and the
    contact count stays within the engine's cap."""
    _, _, m = bench_model
    o = oracle_lib.Oracle(m.to_blob(), "f64")
    o.ctrl[:42] = 0.0          # fold the legs toward zero angles
    o.qpos[2] = 0.3
    o.step(1200)
    i = o.ints()
    assert i["overflow"] == 0 and 1 <= i["ncon"] <= 48
    assert np.isfinite(o.qpos).all()
    per_geom = np.bincount(i["con_geom"], minlength=55)
    assert per_geom.max() <= 4
