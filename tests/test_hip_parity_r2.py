"""Round-2 parity tests of the HIP engine against the CPU oracle on MI355X: the holes VERDICT r1 listed.

* contact geom ids at the boundary (``NMF_CONTACT_GEOM``): bit-equal to the float32 AND float64 oracle's ``con_geom``;
* the whole contact-sensor block (found, force, torque, position, normal, tangent);
* worlds sampled from the full 4096 batch compared with the float64 oracle (not only batch properties);
* BASELINE configs 4 / 5 at their real per-GPU sizes (4096 gapped / blocks, 1024 mixed + odor + gait adhesion);
* the kernel's converged solution against the oracle's MuJoCo-documented solver variant (tolerance tests only, line
  search to its fixed point) — a stopping-rule-independent anchor;
* every named engine semantic (``EngineSemantics``) read identically by kernel and oracle;
* adhesion through the adhesion segment's own geom on the fused LEGS_ACTIVE_ONLY skeleton.

Tolerances as in tests/test_hip_parity.py (float32 engine vs float64 oracle): one step from identical states —
``qacc`` 2e-3 of max |qacc|, ``qpos`` 1e-5, integer quantities bit-exact.
"""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_mod():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    return torch


def _push_state(sim, torch, qpos, qvel, ctrl, ws):
    sim.field("qpos")[:] = torch.as_tensor(np.asarray(qpos), dtype=torch.float32, device=sim.device)
    sim.field("qvel")[:] = torch.as_tensor(np.asarray(qvel), dtype=torch.float32, device=sim.device)
    sim.field("ctrl")[:] = torch.as_tensor(np.asarray(ctrl), dtype=torch.float32, device=sim.device)
    sim.field("qacc_warmstart")[:] = torch.as_tensor(np.asarray(ws), dtype=torch.float32, device=sim.device)


def _walking_states(oracle_lib, blob, key_ctrl, n, seed, settle=400, nact=42, press=0.0):
    """Contact-rich states along a driven trajectory of the float64 oracle (as tests/test_hip_parity._sample_states);
    ``press`` > 0 lowers the root of every third state by that much (tarsal hulls dig in: several contacts per leg)."""
    rng = np.random.default_rng(seed)
    o = oracle_lib.Oracle(blob, "f64")
    o.ctrl[nact:] = 1.0
    o.step(settle)
    states = []
    for k in range(n):
        o.ctrl[:nact] = key_ctrl[:nact] + rng.normal(0, 0.25, nact)
        o.step(60)
        qvel = o.qvel.copy() + rng.normal(0, 0.5, o.nv) * (k % 2)
        qpos = o.qpos.copy()
        if press > 0 and k % 3 == 2:
            qpos[2] -= press
        states.append((qpos, qvel, o.ctrl.copy(), o.arr("qacc_warmstart").copy()))
    return states


def _step_oracle_from(oracle_lib, blob, precision, state, mode="shared"):
    o = oracle_lib.Oracle(blob, precision)
    o.set_solver_mode(mode)
    o.qpos[:] = state[0]; o.qvel[:] = state[1]; o.ctrl[:] = state[2]; o.arr("qacc_warmstart")[:] = state[3]
    o.step(1)
    return o


def test_contact_geom_ids_and_full_sensor_block(torch_mod, bench_model, oracle_lib):
    """SURVEY a6 'ncon, pair ids (bit-exact target)' and a10: one step from 16 contact-rich states."""
    torch = torch_mod
    from flygym_amd import HIPSimulation

    fly, world, m = bench_model
    n = 16
    sim = HIPSimulation(world, n_worlds=n, device=0)
    blob = sim.model.to_blob()
    states = _walking_states(oracle_lib, blob, sim.model["key_ctrl"], n, seed=21, press=0.08)
    _push_state(sim, torch, *[np.stack([s[i] for s in states]) for i in range(4)])
    sim.step(1)
    torch.cuda.synchronize()
    geom = sim.field("contact_geom").cpu().numpy()
    stats = sim.field("stats").cpu().numpy()
    sens = sim.field("sensordata").cpu().numpy().reshape(n, 6, 16)
    assert geom.shape == (n, 48)
    legs_seen, multi = 0, 0
    for w, st in enumerate(states):
        o32 = _step_oracle_from(oracle_lib, blob, "f32", st)
        o64 = _step_oracle_from(oracle_lib, blob, "f64", st)
        ncon = int(stats[w, 0])
        ids = geom[w, :ncon].astype(np.int64)
        assert ncon == o32.ints()["ncon"] == o64.ints()["ncon"]
        np.testing.assert_array_equal(geom[w, :ncon], ids)                      # integers stored exactly
        assert ids.tolist() == o32.ints()["con_geom"] == o64.ints()["con_geom"]   # pair ids, bit-exact, in order
        assert (geom[w, ncon:] == -1).all()
        assert (np.diff(sim.model["geom_body"][ids]) >= 0).all()                # sorted by body
        so = o64.arr("sensordata").reshape(6, 16)
        sh = sens[w]
        np.testing.assert_array_equal(sh[:, 0], so[:, 0])                       # found
        fmax, tmax = np.abs(so[:, 1:4]).max(), np.abs(so[:, 4:7]).max()
        np.testing.assert_allclose(sh[:, 1:4], so[:, 1:4], rtol=5e-3, atol=5e-3 * fmax)       # net force
        # torque about the force-weighted centroid: a difference of nearly equal moments; bounded by the force error
        # times the contact patch size (~0.1 mm)
        np.testing.assert_allclose(sh[:, 4:7], so[:, 4:7], rtol=2e-2, atol=max(2e-2 * tmax, 5e-3 * fmax * 0.1))
        np.testing.assert_allclose(sh[:, 7:10], so[:, 7:10], atol=1e-4)          # position
        np.testing.assert_array_equal(sh[:, 10:13], so[:, 10:13].astype(np.float32))     # normal (exact: plane constants)
        np.testing.assert_array_equal(sh[:, 13:16], so[:, 13:16].astype(np.float32))     # first tangent
        legs_seen += int((so[:, 0] > 0).sum())
        multi += int((so[:, 0] > 1).sum())
    assert legs_seen >= 40 and multi >= 6        # the block was exercised, including multi-contact legs (non-zero torque)


def test_worlds_of_the_full_batch_follow_the_oracle(torch_mod, bench_model, oracle_lib):
    """4096 worlds (BASELINE config 2 size) under the reference benchmark's kinematic replay: 32 worlds drawn at
    random from the batch are compared with the float64 oracle stepping the same partition of the table — state after
    450 steps from reset, contact counts and geom ids of the final step."""
    torch = torch_mod
    from flygym_amd import HIPSimulation
    from flygym_amd.compose import ActuatorType
    from flygym_amd.replay import ReplayTargetData

    fly, world, _ = bench_model
    n = 4096
    order = fly.get_actuated_jointdofs_order(ActuatorType.POSITION)
    table_np = ReplayTargetData(1e-4, order).make_target_angles_all_worlds(n, 1000)
    sim = HIPSimulation(world, n_worlds=n, device=0)
    ids = sim.replay_ids(fly.name)
    sim.set_leg_adhesion_states(fly.name, np.ones((n, 6), dtype=np.float32))
    sim.step(300)
    sim.step_replay(torch.as_tensor(table_np, device=sim.device), ids, 0, 150)
    torch.cuda.synchronize()
    qpos = sim.field("qpos").cpu().numpy()
    stats = sim.field("stats").cpu().numpy()
    geom = sim.field("contact_geom").cpu().numpy()
    # the control (round-4 advisor finding): the same rollout on the primal Newton loop — round 3's solver, whose bars were 85 %
    # followed — judged on the same picks below: the contact-space solve must follow at least as many worlds as it does
    ctl = HIPSimulation(world, n_worlds=n, device=0, _options=dict(solver="primal"))
    ctl.set_leg_adhesion_states(fly.name, np.ones((n, 6), dtype=np.float32))
    ctl.step(300)
    ctl.step_replay(torch.as_tensor(table_np, device=ctl.device), ids, 0, 150)
    torch.cuda.synchronize()
    qpos_ctl = ctl.field("qpos").cpu().numpy()
    assert ctl.batch_info()["solver_option_bits"] == 1 and ctl.get_solver_exits()["contact_space"] == 0
    del ctl
    picks = np.random.default_rng(2).choice(n, size=32, replace=False)
    blob = sim.model.to_blob()
    bases = {}
    for prec in ("f64", "f32"):
        bases[prec] = oracle_lib.Oracle(blob, prec)
        bases[prec].ctrl[42:] = 1.0
        bases[prec].step(300)
    errs, errs64, same_contacts, spread, errs_ctl = [], [], [], [], []
    rng = np.random.default_rng(5)
    for w in picks:
        ref = {}
        for prec in ("f64", "f32"):
            ref[prec] = bases[prec].clone_data()
            ref[prec].step_replay(table_np[w], np.arange(42), 0, 150)
        e64, e32 = np.abs(qpos[w] - ref["f64"].qpos).max(), np.abs(qpos[w] - ref["f32"].qpos).max()
        errs64.append(e64); errs.append(min(e64, e32))
        errs_ctl.append(min(np.abs(qpos_ctl[w] - ref["f64"].qpos).max(), np.abs(qpos_ctl[w] - ref["f32"].qpos).max()))
        # how far the float64 oracle itself lands from its own trajectory when its joint angles are jittered by 3e-7 every
        # five steps (what float32 arithmetic does to a state all along): the sensitivity of this clip partition.  A contact
        # that meets its margin within that jitter enters the list a step earlier or later, and the stiff contact force kicks
        # the trajectory 1e-4 .. 1e-3 away.
        sp = 0.0
        for _ in range(4):
            o = bases["f64"].clone_data()
            for k0 in range(0, 150, 5):
                o.qpos[7:] += 3e-7 * rng.standard_normal(o.nq - 7)
                o.step_replay(table_np[w], np.arange(42), k0, 5)
            sp = max(sp, float(np.abs(o.qpos - ref["f64"].qpos).max()))
        spread.append(sp)
        nc = int(stats[w, 0])
        same_contacts.append(any(nc == r.ints()["ncon"] and geom[w, :nc].astype(int).tolist() == r.ints()["con_geom"]
                                 for r in ref.values()))
    errs, errs64 = np.array(errs), np.array(errs64)
    # 450 steps of a contact-rich rollout.  Most clip partitions are followed to float32 rounding (1e-6); in a few a
    # contact crosses its margin one step apart in float32 and float64 — the float32 ORACLE then also leaves the float64
    # one by 1e-5 .. 4e-4 (partitions 13 and 17 of the clip) — and the stiff contact kicks the trajectories apart.  So:
    # the bulk tight, against whichever oracle the engine's rounding happens to follow; every world bounded.
    # Round 4: no percentage for the partitions that diverge.  A world is followed (< 5e-5) or its partition is one where the
    # float64 oracle itself, jittered at float32's scale, lands as far off as the engine does (chaotic over this horizon: which of those
    # partitions a given build leaves depends on its rounding — the contact-space solve of round 4, more accurate per step
    # than the primal loop, leaves other ones than round 3's kernel did).
    spread = np.array(spread)
    explained = (errs < 5e-5) | (errs < 5.0 * spread)
    print(f"followed {int((errs < 5e-5).sum())} / {len(errs)}, chaotic partitions {int(((errs >= 5e-5) & explained).sum())}, unexplained {int((~explained).sum())}")
    assert explained.all(), (np.sort(errs)[-6:], spread[np.argsort(errs)[-6:]])
    assert (errs < 5e-5).mean() >= 0.6
    followed, followed_ctl = int((errs < 5e-5).sum()), int((np.array(errs_ctl) < 5e-5).sum())
    print(f"control (primal loop) followed {followed_ctl} / {len(errs)}")
    from ledger import report
    report("worlds_of_the_full_batch_follow_the_oracle", followed=followed, followed_primal_control=followed_ctl, picks=len(errs),
           chaotic_partitions=int(((errs >= 5e-5) & explained).sum()), unexplained=int((~explained).sum()), median_vs_f64=float(np.median(errs64)),
           worst_vs_f64=float(errs64.max()))
    assert followed >= followed_ctl - 2, (followed, followed_ctl)         # not worse than round 3's solver on the same picks (two worlds of slack: which chaotic partition a build leaves is its rounding's)
    assert np.median(errs64) < 5e-6 and errs64.max() < 5e-3, np.sort(errs64)[-6:]
    same_contacts = np.array(same_contacts)
    assert same_contacts[errs < 5e-5].mean() >= 0.95 and same_contacts.mean() >= 0.7      # the followed worlds end on the oracle's contact list
    assert len({int(w) % 20 for w in picks}) >= 12          # the sample spans the clip partitions


@pytest.mark.parametrize("terrain", ["GappedTerrainWorld", "BlocksTerrainWorld"])
def test_config4_terrain_at_full_size(torch_mod, terrain):
    """BASELINE config 4 per GPU: 4096 flies CPG-walking over gapped / blocks terrain for 0.15 s — finite, no contact
    overflow, solver below its cap, every fly upright and moving forward, deterministic across two runs."""
    torch = torch_mod
    import flygym_amd.compose as C
    from flygym_amd import HIPSimulation, make_model
    from flygym_amd.controllers import TripodCPG
    from flygym_amd.utils.math import Rotation3D

    n = 4096
    finals = []
    for rep in range(2):
        fly, _, _ = make_model()
        world = getattr(C, terrain)()
        world.add_fly(fly, (0, 0, 0.8), Rotation3D("quat", (1, 0, 0, 0)))
        sim = HIPSimulation(world, n_worlds=n, device=0)
        cpg = TripodCPG(fly.get_actuated_jointdofs_order("position"), 1e-4)
        table = cpg.targets(n, 2500, device=sim.device)
        ids = sim.replay_ids(fly.name)
        sim.set_leg_adhesion_states(fly.name, np.ones((n, 6), dtype=np.float32))
        sim.warmup()
        x0 = sim.field("qpos")[:, 0].clone()
        s0 = sim.field("stats_sum").clone()
        worst = 0
        for tick in range(30):
            sim.step_replay(table, ids, 50 * tick, 50)
            worst = max(worst, int(sim.field("stats")[:, 1].max().item()))
        torch.cuda.synchronize()
        q = sim.field("qpos")
        ds = (sim.field("stats_sum") - s0).double().sum(dim=0)
        assert bool(torch.isfinite(q).all()) and bool(torch.isfinite(sim.field("qvel")).all())
        assert float(ds[0]) == n * 1500 and float(ds[3]) == 0.0          # every step counted, no overflow at any step
        assert float(ds[1] / ds[0]) > 3.0 and float(ds[2] / ds[0]) > 1.0  # contact-rich stepping
        assert worst < 60
        up = 1 - 2 * (q[:, 4] ** 2 + q[:, 5] ** 2)
        assert float(up.min()) > 0.5 and float((up > 0.9).float().mean()) > 0.97      # a stumble on a block edge, no fall (~1 % tilt past 0.9 at some point)
        # 1.8 gait cycles forward.  (Round 2, tops only: > 0.15 mm.  With the cells' side faces a tarsus that drops into a
        # gap or meets a raised square is held by the wall instead of sliding through it: the open-loop gait, which does
        # not know the terrain, gets 0.06 / 0.13 mm on the gapped / blocks worlds.)
        assert float((q[:, 0] - x0).median()) > 0.03
        finals.append(q.clone())
        del sim
    assert torch.equal(finals[0], finals[1])


def test_config5_mixed_terrain_odor_adhesion_at_full_size(torch_mod):
    """BASELINE config 5: 1024 flies on mixed terrain with gait-driven leg adhesion and the four odor sensors read
    every control tick."""
    torch = torch_mod
    import flygym_amd.compose as C
    from flygym_amd import HIPSimulation, make_model
    from flygym_amd.controllers import TripodCPG
    from flygym_amd.sensors import OdorSensors
    from flygym_amd.utils.math import Rotation3D

    n = 1024
    fly, _, _ = make_model()
    world = C.MixedTerrainWorld()
    world.add_fly(fly, (0, 0, 0.8), Rotation3D("quat", (1, 0, 0, 0)))
    sim = HIPSimulation(world, n_worlds=n, device=0)
    cpg = TripodCPG(fly.get_actuated_jointdofs_order("position"), 1e-4)
    table = cpg.targets(n, 2500, device=sim.device, adhesion=(cpg.stance_bins(sim.model, fly), 20.0, 1.0))
    ids = sim.replay_ids(fly.name, with_adhesion=True)
    rng = np.random.default_rng(0)
    src = rng.uniform(-20, 20, (3, 3)); src[:, 2] = rng.uniform(0.5, 3.0, 3)
    odor = OdorSensors(sim, fly.name, src, rng.uniform(0.1, 1.0, (3, 2)))
    sim.set_leg_adhesion_states(fly.name, np.ones((n, 6), dtype=np.float32))
    sim.warmup()
    x0 = sim.field("qpos")[:, 0].clone()
    s0 = sim.field("stats_sum").clone()
    readings = []
    for tick in range(40):
        sim.step_replay(table, ids, 50 * tick, 50)
        readings.append(odor.get_odor_intensities().clone())
    torch.cuda.synchronize()
    q = sim.field("qpos")
    ds = (sim.field("stats_sum") - s0).double().sum(dim=0)
    assert bool(torch.isfinite(q).all())
    assert float(ds[0]) == n * 2000 and float(ds[3]) == 0.0
    assert float(ds[1] / ds[0]) > 3.0 and float(ds[2] / ds[0]) > 1.0
    r = torch.stack(readings)
    assert tuple(r.shape) == (40, n, 2, 4) and bool(torch.isfinite(r).all()) and float(r.min()) > 0.0
    assert float((r[-1] - r[0]).abs().max()) > 0.0                         # the flies moved through the plume
    assert float((q[:, 0] - x0).median()) > 0.2
    up = 1 - 2 * (q[:, 4] ** 2 + q[:, 5] ** 2)
    assert float(up.min()) > 0.5 and float((up > 0.9).float().mean()) > 0.97


def test_kernel_solution_matches_the_documented_solver_variant(torch_mod, bench_model, oracle_lib):
    """The oracle's 'documented' mode keeps MuJoCo's stopping rules only (gradient / improvement against `tolerance`,
    line search iterated to its fixed point) and is never edited together with the kernel.  The kernel's *solution*
    — not its iteration count — must match it: one step from 12 contact-rich states."""
    torch = torch_mod
    from flygym_amd import HIPSimulation

    fly, world, _ = bench_model
    n = 12
    sim = HIPSimulation(world, n_worlds=n, device=0)
    blob = sim.model.to_blob()
    states = _walking_states(oracle_lib, blob, sim.model["key_ctrl"], n, seed=33)
    _push_state(sim, torch, *[np.stack([s[i] for s in states]) for i in range(4)])
    sim.step(1)
    torch.cuda.synchronize()
    qacc = sim.field("qacc").cpu().numpy()
    qpos = sim.field("qpos").cpu().numpy()
    ncon = sim.field("stats").cpu().numpy()[:, 0]
    for w, st in enumerate(states):
        od = _step_oracle_from(oracle_lib, blob, "f64", st, mode="documented")
        osh = _step_oracle_from(oracle_lib, blob, "f64", st, mode="shared")
        scale = np.abs(od.arr("qacc")).max()
        # the two oracle variants converge to the same optimum (strictly convex cost) ...
        assert np.abs(od.arr("qacc") - osh.arr("qacc")).max() < 1e-6 * scale
        assert od.ints()["solver_iter"] >= osh.ints()["solver_iter"]
        # ... and so does the kernel
        assert int(ncon[w]) == od.ints()["ncon"]
        assert np.abs(qacc[w] - od.arr("qacc")).max() < 2e-3 * scale
        assert np.abs(qpos[w] - od.qpos).max() < 1e-5


@pytest.mark.parametrize("flag", ["pyramid_R=plain", "sensor_frame=contact", "max_hull_contacts=1",
                                  "max_hull_contacts=2", "invweight0=fused_body", "mesh_inertia=convex",
                                  "capsule_fit=aabb"])
def test_named_semantics_are_read_by_kernel_and_oracle_alike(torch_mod, oracle_lib, flag):
    """Each low-confidence MuJoCo semantic of SURVEY Appendix A is a switch on ``world.semantics``; flipping one must
    change kernel and oracle together (one step from contact-rich states + a 200-step rollout)."""
    torch = torch_mod
    from flygym_amd import HIPSimulation, make_model

    key, val = flag.split("=")
    fly, world, _ = make_model()
    setattr(world.semantics, key, int(val) if val.isdigit() else val)
    fly0, world0, _ = make_model()
    n = 6
    sim = HIPSimulation(world, n_worlds=n, device=0)
    blob = sim.model.to_blob()
    assert blob != world0.compile_model().to_blob()                       # the switch reaches the compiled model
    states = _walking_states(oracle_lib, blob, sim.model["key_ctrl"], n, seed=5)
    _push_state(sim, torch, *[np.stack([s[i] for s in states]) for i in range(4)])
    sim.step(1)
    torch.cuda.synchronize()
    qacc = sim.field("qacc").cpu().numpy()
    stats = sim.field("stats").cpu().numpy()
    geom = sim.field("contact_geom").cpu().numpy()
    sens = sim.field("sensordata").cpu().numpy().reshape(n, 6, 16)
    for w, st in enumerate(states):
        o = _step_oracle_from(oracle_lib, blob, "f64", st)
        nc = int(stats[w, 0])
        assert nc == o.ints()["ncon"] and geom[w, :nc].astype(int).tolist() == o.ints()["con_geom"]
        scale = np.abs(o.arr("qacc")).max()
        assert np.abs(qacc[w] - o.arr("qacc")).max() < 2e-3 * scale
        so = o.arr("sensordata").reshape(6, 16)
        np.testing.assert_array_equal(sens[w][:, 0], so[:, 0])
        np.testing.assert_allclose(sens[w][:, 1:4], so[:, 1:4], rtol=5e-3, atol=5e-3 * np.abs(so[:, 1:4]).max())
    sim.reset()
    sim.set_leg_adhesion_states(fly.name, np.ones((n, 6), dtype=np.float32))
    o = oracle_lib.Oracle(blob, "f64")
    o.ctrl[42:] = 1.0
    sim.step(200); o.step(200)
    assert np.abs(sim.field("qpos").cpu().numpy() - o.qpos[None]).max() < 5e-5


@pytest.mark.parametrize("mode", ["segment_geom", "fused_body"])
def test_adhesion_through_the_adhesion_segments_geom(torch_mod, oracle_lib, mode):
    """LEGS_ACTIVE_ONLY fuses tarsus1..5; with tarsus1-4 pressed into the ground the adhesion pull
    must go through the tarsus5 contacts only.  HIP vs float64 oracle, one step from the pressed state, both readings."""
    torch = torch_mod
    from flygym_amd import HIPSimulation, make_model
    from flygym_amd.anatomy import JointPreset

    fly, world, _ = make_model(joints_preset=JointPreset.LEGS_ACTIVE_ONLY)
    world.semantics.adhesion_contacts = mode
    sim = HIPSimulation(world, n_worlds=2, device=0)
    blob = sim.model.to_blob()
    o = oracle_lib.Oracle(blob, "f64")
    o.ctrl[42:] = 40.0
    o.step(600)
    o.qpos[2] -= 0.12
    o.qvel[:] = 0.0
    st = (o.qpos.copy(), o.qvel.copy(), o.ctrl.copy(), o.arr("qacc_warmstart").copy())
    _push_state(sim, torch, *[np.stack([s, s]) for s in st])
    sim.step(1)
    torch.cuda.synchronize()
    ref = _step_oracle_from(oracle_lib, blob, "f64", st)
    nc = int(sim.field("stats")[0, 0].item())
    ids = sim.field("contact_geom")[0, :nc].cpu().numpy().astype(int)
    assert nc == ref.ints()["ncon"] and ids.tolist() == ref.ints()["con_geom"]
    tarsus5 = set(int(g) for g in sim.model["act_geom"][42:])
    assert sum(1 for g in ids if g not in tarsus5 and sim.model["geom_body"][g] in set(sim.model["act_trn"][42:])) >= 4
    qa = sim.field("qacc").cpu().numpy()[0]
    assert np.abs(qa - ref.arr("qacc")).max() < 2e-3 * np.abs(ref.arr("qacc")).max()
    np.testing.assert_allclose(sim.field("actuator_force").cpu().numpy()[0], ref.arr("actuator_force"), rtol=1e-4, atol=1e-4)
    # the two readings give different accelerations in this state (the force of 40 per leg is shared differently)
    other = "fused_body" if mode == "segment_geom" else "segment_geom"
    fly2, world2, _ = make_model(joints_preset=JointPreset.LEGS_ACTIVE_ONLY)
    world2.semantics.adhesion_contacts = other
    ref2 = _step_oracle_from(oracle_lib, world2.compile_model().to_blob(), "f64", st)
    assert np.abs(ref2.arr("qacc") - ref.arr("qacc")).max() > 1e-2 * np.abs(ref.arr("qacc")).max()


def test_stats_sum_and_step_replay_validation(torch_mod, bench_model):
    """NMF_STATS_SUM accumulates (steps, contacts, solver iterations, overflow steps, and one counter per bit of the solve
    report: how each step's constraint solve ended) over every step of every launch; step_replay refuses tables it could only
    read as garbage."""
    torch = torch_mod
    from flygym_amd import HIPSimulation

    fly, world, _ = bench_model
    n = 5
    a = HIPSimulation(world, n_worlds=n, device=0)
    b = HIPSimulation(world, n_worlds=n, device=0)
    for s in (a, b):
        s.set_leg_adhesion_states(fly.name, np.ones((n, 6), dtype=np.float32))
    a.step(400)
    acc = torch.zeros((n, 16), device=a.device)
    for _ in range(400):
        b.step(1)
        st = b.field("stats")
        acc[:, 0] += 1; acc[:, 1] += st[:, 0]; acc[:, 2] += st[:, 1]; acc[:, 3] += st[:, 2]
        bits = st[:, 4].int()
        for k in range(12):
            acc[:, 4 + k] += ((bits >> k) & 1).float()
    torch.cuda.synchronize()
    assert torch.equal(a.field("qpos"), b.field("qpos"))
    assert a.field("stats_sum").dtype == torch.int32                 # uint32 counters behind the field pointer
    assert torch.equal(a.field("stats_sum").float(), acc) and torch.equal(b.field("stats_sum").float(), acc)
    assert float(acc[:, 1].min()) > 100          # landed within the 400 steps
    # every step ended exactly one way: in contact space, on the primal loop, or without a contact; a contact-space solve one of five
    assert torch.equal(acc[:, 4] + acc[:, 10] + acc[:, 14], acc[:, 0]) and torch.equal(acc[:, 5:10].sum(dim=1), acc[:, 4])
    a.reset()
    assert float(a.field("stats_sum").abs().max()) == 0.0 and bool((a.field("contact_geom") == -1).all())
    ids = a.replay_ids(fly.name)
    good = torch.zeros((n, 10, 42), device=a.device)
    a.step_replay(good, ids, 0, 2)
    for bad_table, bad_ids in [(good.double(), ids), (good.cpu(), ids), (good[:, :, :41], ids), (good[:4], ids),
                               (good.transpose(0, 1), ids), (good, ids.long()), (good.cpu().numpy(), ids)]:
        with pytest.raises(ValueError):
            a.step_replay(bad_table, bad_ids, 0, 2)


def test_simulation_on_a_device_other_than_the_current_one_is_guarded(torch_mod, bench_model):
    """the ABI sets the batch's device itself; a HIPSimulation built for device 0 keeps working whatever
    torch's current device is (only checkable with one GPU as a no-op guard + a clear error for a missing device)."""
    torch = torch_mod
    from flygym_amd import HIPSimulation, _native

    fly, world, _ = bench_model
    sim = HIPSimulation(world, n_worlds=2, device=0)
    sim.step(3)
    assert sim.time == pytest.approx(3e-4)
    if torch.cuda.device_count() == 1:
        with pytest.raises(_native.NativeError):
            HIPSimulation(world, n_worlds=2, device=7)


def test_replay_table_resampled_on_the_device(torch_mod, bench_model):
    """MotionSnippet's Savitzky-Golay + cubic resample on the GPU (nmf_replay_resample) against the
    reference pipeline's table (scipy on the host; tests/golden/replay_42.npz pins it to the reference): float32 values
    bit-identical but for a few that are one ulp off, and the per-world tables built from it on the device equal the
    host-built ones."""
    torch = torch_mod
    from pathlib import Path
    from flygym_amd.replay import MotionSnippet, ReplayTargetData

    fly, _, _ = bench_model
    order = fly.get_actuated_jointdofs_order("position")
    ms = MotionSnippet()
    dev = ms.get_joint_angles_device(1e-4, order, "cuda:0")
    host = ms.get_joint_angles(1e-4, order).astype(np.float32)
    got = dev.cpu().numpy()
    assert got.shape == host.shape == (20000, 42) and got.dtype == np.float32
    neq = got.view(np.uint32) != host.view(np.uint32)
    assert neq.mean() < 1e-3, f"{int(neq.sum())} of {neq.size} values differ"
    assert np.abs(got - host).max() <= 2.4e-7
    gold = np.load(Path(__file__).parent / "golden" / "replay_42.npz")
    assert (got[:2000].view(np.uint32) == gold["head"].view(np.uint32)).mean() > 0.999
    a = ReplayTargetData(1e-4, order, device="cuda:0").make_target_angles_all_worlds(45, 1000, first_world=3)
    b = ReplayTargetData(1e-4, order).make_target_angles_all_worlds(45, 1000, first_world=3)
    assert tuple(a.shape) == (45, 1000, 42) and a.dtype == torch.float32 and a.is_contiguous()
    assert np.abs(a.cpu().numpy() - b).max() <= 2.4e-7
    assert torch.equal(a[20], a[0]) and not torch.equal(a[1], a[0])          # world w <- partition w % 20
