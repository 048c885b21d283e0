"""Tripod CPG (BASELINE config 2 control; build-defined, see flygym_amd/controllers.py)."""

import numpy as np

from flygym_amd.anatomy import LEGS
from flygym_amd.controllers import TRIPOD_PHASE_BIAS, TripodCPG


def test_tripod_cpg_tables(bench_model, oracle_lib):
    fly, _, m = bench_model
    order = fly.get_actuated_jointdofs_order("position")
    cpg = TripodCPG(order, 1e-4)
    ph = cpg.phases(8, 4)
    legs = {leg: i for i, leg in enumerate(LEGS)}
    for a in ("lf", "rm", "lh"):                        # tripod A in phase, tripod B half a cycle away
        assert ph[0, 0, legs[a]] == ph[0, 0, legs["lf"]]
    for b in ("rf", "lm", "rh"):
        assert np.isclose((ph[0, 0, legs[b]] - ph[0, 0, legs["lf"]]) % (2 * np.pi), np.pi)
    assert np.isclose(ph[1, 0, 0] - ph[0, 0, 0], 2 * np.pi / 8)             # per-world offset 2 pi w / n
    assert np.isclose(ph[0, 1, 0] - ph[0, 0, 0], 2 * np.pi * 12.0 * 1e-4)   # 12 Hz
    t = cpg.targets(3, 5000)
    assert t.shape == (3, 5000, 42) and t.dtype == np.float32
    np.testing.assert_allclose(t[:, :2500], t[:, 2500:], atol=2e-6)         # 3 cycles = 2500 steps: seamless wrap
    assert np.abs(np.diff(t, axis=1)).max() < 0.02                           # smooth targets
    shard = cpg.targets(2, 100, first_world=4, total_worlds=8)
    np.testing.assert_array_equal(shard, cpg.targets(8, 100)[4:6])          # multi-GPU shards see their global phase
    # the gait moves the fly forward on the CPU oracle
    o = oracle_lib.Oracle(m.to_blob(), "f64")
    o.ctrl[42:] = 1.0
    o.step(500)
    x0 = o.qpos[0]
    o.step_replay(t[0], np.arange(42), 0, 2500)
    assert o.qpos[0] - x0 > 1.0 and np.isfinite(o.qpos).all()


def test_cpg_driven_adhesion_columns(bench_model, oracle_lib):
    """BASELINE config 5: adhesion follows the gait.  Stance = claw near its lowest point of the step cycle."""
    import torch

    fly, _, m = bench_model
    order = fly.get_actuated_jointdofs_order("position")
    cpg = TripodCPG(order, 1e-4)
    stance = cpg.stance_bins(m, fly)
    assert stance.shape == (256, 6) and stance.dtype == bool
    frac = stance.mean(axis=0)
    assert ((frac > 0.35) & (frac < 0.75)).all()                       # every leg spends a good part of the cycle on the ground
    for leg in range(6):                                              # one contiguous stance phase per cycle
        assert (np.roll(stance[:, leg], 1) != stance[:, leg]).sum() <= 4
    legs = {leg: i for i, leg in enumerate(LEGS)}
    t = cpg.targets(4, 2500, adhesion=(stance, 20.0, 1.0))
    assert t.shape == (4, 2500, 48)
    np.testing.assert_array_equal(t[..., :42], cpg.targets(4, 2500))
    assert set(np.unique(t[..., 42:])) == {1.0, 20.0}
    # tripod A and tripod B alternate: lf and rf are never both in swing for long, and their stance is out of phase
    both = (t[0, :, 42 + legs["lf"]] == 20.0) & (t[0, :, 42 + legs["rf"]] == 20.0)
    assert both.mean() < 0.45
    # the torch builder (used on the GPU) gives the same table
    tt = cpg.targets(4, 2500, device="cpu", adhesion=(stance, 20.0, 1.0)).numpy()
    np.testing.assert_allclose(tt[..., :42], t[..., :42], atol=2e-6)
    assert (tt[..., 42:] != t[..., 42:]).mean() < 1e-3              # bin edges may round differently
    # walking with gait-driven adhesion on the CPU oracle: stays finite and moves forward
    o = oracle_lib.Oracle(m.to_blob(), "f64")
    o.ctrl[42:] = 1.0
    o.step(500)
    x0 = o.qpos[0]
    pos_ids = [i for i, a in enumerate(fly.actuators) if a["kind"] == "position"]
    adh_ids = [i for i, a in enumerate(fly.actuators) if a["kind"] == "adhesion"]
    o.step_replay(t[0], np.array(pos_ids + adh_ids), 0, 2500)
    assert np.isfinite(o.qpos).all() and o.qpos[0] - x0 > 0.5


def test_cpg_holds_dofs_the_clip_does_not_have():
    """ALL_POSSIBLE actuates all three axes of every leg joint; the walking clip (reference
    ``assets/behavior/single_steps_untethered.npz`` behind ``MotionSnippet``) has the 42 biological ones.  The CPG's table
    drives those exactly as for LEGS_ONLY and holds the others at zero, their neutral angle."""
    from flygym_amd import make_model
    from flygym_amd.compose import ActuatorType
    from flygym_amd.controllers import TripodCPG
    from flygym_amd.replay import MotionSnippet

    tables, orders = {}, {}
    for preset in ("legs_only", "all_possible"):
        fly = make_model(joints_preset=preset)[0]
        orders[preset] = fly.get_actuated_jointdofs_order(ActuatorType.POSITION)
        tables[preset] = TripodCPG(orders[preset], 1e-4).targets(2, 400)
    key = lambda d: (d.child.pos, d.parent.link, d.child.link, d.axis.value)
    col = {key(d): i for i, d in enumerate(orders["legs_only"])}
    clip_dofs = MotionSnippet().dofs_per_leg
    assert len(orders["legs_only"]) == 42 and len(orders["all_possible"]) == 72
    n_held = 0
    for j, d in enumerate(orders["all_possible"]):
        if (d.parent.link, d.child.link, d.axis.value) in clip_dofs:
            np.testing.assert_array_equal(tables["all_possible"][:, :, j], tables["legs_only"][:, :, col[key(d)]])
        else:
            assert not tables["all_possible"][:, :, j].any()
            n_held += 1
    assert n_held == 30
