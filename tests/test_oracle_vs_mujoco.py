"""The CPU oracle against REAL MuJoCo outputs (tests/golden/mujoco_golden.npz, written by
tests/golden/make_mujoco_golden.py on a box that has mujoco 3.6.0 and the reference installed).

The dump cannot be produced in the build container, so the pinned tests SKIP until it exists; the
harness itself (tests/mujoco_compare.py) is exercised below on dumps the oracle writes under known semantics: it must
recognise which ``EngineSemantics`` switches a dump was made with — that is the mechanism by which a mismatch with
MuJoCo becomes a flag flip.
"""

from pathlib import Path

import numpy as np
import pytest

import mujoco_compare as mc

GOLD = Path(__file__).parent / "golden" / "mujoco_golden.npz"
needs_dump = pytest.mark.skipif(not GOLD.exists(), reason="tests/golden/mujoco_golden.npz absent: run tests/golden/"
                                "make_mujoco_golden.py where mujoco==3.6.0 and the reference are importable")


@needs_dump
def test_compile_time_constants_match_mujoco(bench_model):
    """Masses, inertial frames, inertias and invweight0 per named segment; mean inertia; pair parameters."""
    dump = np.load(GOLD)
    _, _, m = bench_model
    names = [str(n) for n in dump["model/body_names"]]
    seg_names = m.meta["seg_names"]
    # bodies MuJoCo kept after fusestatic that carry a hinge here: masses of the merged bodies must agree
    dyn_of_seg = m["seg_body"]
    for b in range(1, m.nb):
        segs = [s for s in range(len(seg_names)) if dyn_of_seg[s] == b]
        moving = [s for s in segs if seg_names[s] in names]
        assert moving, f"dynamic body {b} has no MuJoCo counterpart"
        mj_mass = sum(float(dump["model/body_mass"][names.index(seg_names[s])]) for s in moving)
        assert m["body_mass"][b] == pytest.approx(mj_mass, rel=1e-6), seg_names[moving[0]]
    assert float(m["stat_meaninertia"][0]) == pytest.approx(float(dump["model/meaninertia"][0]), rel=1e-4)
    np.testing.assert_allclose(m["pair_margin"], dump["model/pair_margin"][: m.ng], rtol=1e-12)
    np.testing.assert_allclose(m["pair_solref"], dump["model/pair_solref"][: m.ng], rtol=1e-12)
    np.testing.assert_allclose(m["pair_solimp"], dump["model/pair_solimp"][: m.ng], rtol=1e-9)
    np.testing.assert_allclose(m["pair_friction"], dump["model/pair_friction"][: m.ng], rtol=1e-12)
    for gi, s in enumerate(m["geom_seg"]):
        b = names.index(seg_names[s])
        assert m["geom_invweight0"][gi] == pytest.approx(float(dump["model/body_invweight0"][b, 0]), rel=1e-3), seg_names[s]


@needs_dump
def test_oracle_step_matches_mujoco(oracle_lib):
    """ncon and the contact set exactly; qacc_smooth 1e-6, qacc 1e-4, actuator forces 1e-9, sensor force 1e-3 (relative
    to each quantity's largest magnitude); next qpos 1e-7.  On failure the message names the EngineSemantics
    combination that fits the dump best."""
    dump = np.load(GOLD)
    res = mc.residuals(oracle_lib, dump)
    ok = (res["ncon_mismatch"] == 0 and res["contact_set_mismatch"] == 0 and res["sensor_found"] == 0 and res["qacc_smooth"] < 1e-6
          and res["qacc"] < 1e-4 and res["actuator_force"] < 1e-9 and res["sensor_force"] < 1e-3 and res["next_qpos"] < 1e-7)
    if not ok:
        best = mc.rank_semantics(oracle_lib, dump, include_compile_time=True)[:3]
        pytest.fail(f"default semantics deviate from MuJoCo: {res}\nbest-fitting switches:\n" +
                    "\n".join(f"  score {s:.3g}: {sem} -> {r}" for s, sem, r in best))


@pytest.mark.parametrize("truth", [dict(pyramid_R="plain"), dict(sensor_frame="contact", max_hull_contacts=1), dict()])
def test_harness_recovers_the_semantics_a_dump_was_made_with(oracle_lib, truth):
    """Harness self-test (no MuJoCo): a dump written by the oracle under `truth` is fitted exactly by `truth` and by no
    combination that differs in a switch the states exercise."""
    dump = mc.oracle_as_dump(oracle_lib, n_states=4, **truth)
    exact = mc.residuals(oracle_lib, dump, **truth)
    assert mc.score(exact) < 1e-9, exact
    ranked = mc.rank_semantics(oracle_lib, dump)
    best_score, best_sem, _ = ranked[0]
    assert best_score < 1e-9
    full_truth = {k: truth.get(k, v[0]) for k, v in mc.RUNTIME.items()}
    for key in ("pyramid_R", "sensor_frame"):
        assert best_sem[key] == full_truth[key]
    wrong = [r for r in ranked if r[1]["pyramid_R"] != full_truth["pyramid_R"]]
    assert min(w[0] for w in wrong) > 1e-4              # a wrong regulariser is visible in qacc
    assert set(mc.states_of(dump)[0]) >= {"qpos", "qvel", "ctrl", "qacc_warmstart", "ncon", "con_segment", "qacc", "sensordata"}
