"""Round 5: one contact-space solve for every walking step and every batch size (the Gram matrix of the contact DIRECTIONS,
csrc/nmf_dual.h), the solver's exit accounting, the per-step observation ring.  GPU tests, through the C ABI."""
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = Path(__file__).parent / "golden"


@pytest.fixture(scope="module")
def torch_mod():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    return torch


def _world(kind):
    import flygym_amd.compose as C
    from flygym_amd import make_model
    from flygym_amd.utils.math import Rotation3D

    if kind in ("legs_only", "legs_active_only", "all_biological"):
        fly, world, _ = make_model(joints_preset=kind)
        return fly, world
    fly = make_model(joints_preset="legs_only")[0]
    world = getattr(C, {"blocks": "BlocksTerrainWorld", "mixed": "MixedTerrainWorld"}[kind])()
    world.add_fly(fly, (0.3, 0.2, 0.8), Rotation3D("quat", (1, 0, 0, 0)))
    return fly, world


STATE = ("qpos", "qvel", "qacc_warmstart", "ctrl", "time", "stats_sum", "stats", "sensordata", "seg_xpos", "contact_geom",
         "actuator_force", "qacc")


@pytest.mark.parametrize("kind", ["legs_only", "legs_active_only", "all_biological", "blocks"])
def test_a_world_does_not_depend_on_the_size_of_its_batch(torch_mod, kind):
    """Worlds 0..511 of a 4096-world batch against the same 512 worlds stepped as a batch of their own — and as a batch of 64:
    every state array, the clock, the statistics and the outputs bit for bit after a settle and 600 steps of CPG walking in
    launches of three lengths.  (Rounds 3-4 chose between two LEGS_ONLY kernel flavours by batch size — 12 or 16 contacts in the
    contact-space solve — and a sharded run agreed with the single-GPU run only to solver tolerance.  One kernel per skeleton
    now: the schedule, the residency and the shard plan never change a result.)"""
    torch = torch_mod
    from flygym_amd import HIPSimulation
    from flygym_amd.controllers import TripodCPG

    fly, world = _world(kind)
    big_n = 4096
    table = TripodCPG(fly.get_actuated_jointdofs_order("position"), 1e-4).targets(big_n, 1250, device="cuda:0")

    def run(n):
        sim = HIPSimulation(world, n_worlds=n, device=0)
        ids = sim.replay_ids(fly.name)
        sim.set_leg_adhesion_states(fly.name, np.ones((n, 6), dtype=np.float32))
        tab = table[:n].contiguous()
        sim.step(500)
        cur = 0
        for length, count in ((50, 8), (20, 9), (1, 20)):
            for _ in range(count):
                sim.step_replay(tab, ids, cur, length); cur += length
        torch.cuda.synchronize()
        return {k: sim.field(k).clone() for k in STATE}

    big = run(big_n)
    assert bool(torch.isfinite(big["qpos"]).all())
    assert float(big["stats_sum"][:, 1].float().mean()) / 1100 > 3.0          # the flies stand and walk (contacts per step)
    for n in (512, 64):
        small = run(n)
        for k in STATE:
            assert torch.equal(small[k], big[k][:n]), f"{kind}: worlds 0..{n - 1} of a {big_n}-batch differ from a {n}-batch in {k}"


def test_observation_ring_rows_are_the_per_step_reads(torch_mod):
    """``nmf_step_record``: one fused launch of 40 steps that records the observation block of every step (and of every 5th)
    against 40 one-step launches each followed by ``pack_observations`` and the getters — what the reference's loops do after
    every ``step()`` (reference simulation.py:142-243).  Bit for bit, including the contact sensors and the actuator forces,
    which the kernel otherwise evaluates on a launch's last step only; the final state is the same too."""
    torch = torch_mod
    from flygym_amd import HIPSimulation, make_model
    from flygym_amd.compose.fly import ActuatorType
    from flygym_amd.controllers import TripodCPG

    fly, world, _ = make_model()
    n = 4096
    table = TripodCPG(fly.get_actuated_jointdofs_order("position"), 1e-4).targets(n, 1250, device="cuda:0")
    sims = [HIPSimulation(world, n_worlds=n, device=0) for _ in range(3)]
    ids = sims[0].replay_ids(fly.name)
    for sim in sims:
        sim.set_leg_adhesion_states(fly.name, np.ones((n, 6), dtype=np.float32))
        sim.step(500); sim.step_replay(table, ids, 0, 300)
    a, b, c = sims
    ring = a.step_replay(table, ids, 300, 40, record_every=1)
    ring5 = c.step_replay(table, ids, 300, 40, record_every=5)
    assert tuple(ring.shape) == (40, n, 270) and tuple(ring5.shape) == (8, n, 270)
    row = torch.empty((n, 270), device=b.device)
    touched = 0
    for s in range(40):
        b.step_replay(table, ids, 300 + s, 1)
        b.pack_observations(row)
        assert torch.equal(ring[s], row), f"ring row {s} differs from the per-step read"
        if s % 5 == 4:
            assert torch.equal(ring5[s // 5], row)
        if s == 17:       # the getters of the reference surface read the same numbers
            assert torch.equal(ring[s][:, :66], b.get_joint_angles(fly.name)) and torch.equal(ring[s][:, 66:132], b.get_joint_velocities(fly.name))
            assert torch.equal(ring[s][:, 132:174], b.get_actuator_forces(fly.name, ActuatorType.POSITION))
            act = b.get_ground_contact_info(fly.name)[0]
            assert torch.equal(ring[s][:, 174:].reshape(n, 6, 16)[:, :, 0], act)
        touched += int((row[:, 174::16] > 0).sum().item())
    assert touched > 40 * n          # legs on the ground in every step: the sensor block is not a block of zeros
    torch.cuda.synchronize()
    for k in STATE:
        assert torch.equal(a.field(k), b.field(k)) and torch.equal(c.field(k), b.field(k)), k
    with pytest.raises(ValueError):
        a.step(3, record_every=5)


WORKLOADS = {"config2": ("legs_only", "flat", 1.0), "config4_blocks": ("legs_only", "blocks", 1.0), "config5_mixed_adhesion": ("legs_only", "mixed", 20.0)}


@pytest.mark.parametrize("name", list(WORKLOADS))
def test_how_the_solves_end(torch_mod, name):
    """2 000 steps x 4096 worlds of BASELINE configs 2 / 4 / 5 with the solver's exit accounting on (``NMF_STATS_SUM`` columns
    4..14): every step ends exactly one way; at least 99.9 % of the contact-space solves end on the exact KKT test; whatever
    ends otherwise passed the residual test (what the last elimination's target violates is at most 1e-3 of the rows' residuals)
    or was solved again on the primal loop and counted; no step is left unsolved (finite state, no iteration limit)."""
    torch = torch_mod
    import flygym_amd.compose as C
    from flygym_amd import HIPSimulation, make_model
    from flygym_amd.controllers import TripodCPG
    from flygym_amd.utils.math import Rotation3D

    preset, terrain, adhesion = WORKLOADS[name]
    fly = make_model(joints_preset=preset)[0]
    world = {"flat": C.FlatGroundWorld, "blocks": C.BlocksTerrainWorld, "mixed": C.MixedTerrainWorld}[terrain]()
    world.add_fly(fly, (0, 0, 0.8), Rotation3D("quat", (1, 0, 0, 0)))
    n = 4096
    sim = HIPSimulation(world, n_worlds=n, device=0)
    cpg = TripodCPG(fly.get_actuated_jointdofs_order("position"), 1e-4)
    sim.set_leg_adhesion_states(fly.name, np.ones((n, 6), dtype=np.float32))
    if adhesion > 1.0:       # adhesion driven by the gait: `adhesion` while a leg's phase is a stance phase, else 1 (bench.py --cpg-adhesion)
        table = cpg.targets(n, 1250, device="cuda:0", adhesion=(cpg.stance_bins(sim.model, fly), adhesion, 1.0))
        ids = sim.replay_ids(fly.name, with_adhesion=True)
    else:
        table = cpg.targets(n, 1250, device="cuda:0")
        ids = sim.replay_ids(fly.name)
    sim.step(500)
    s0 = sim.field("stats_sum").clone()
    cur = 0
    for _ in range(40):
        sim.step_replay(table, ids, cur, 50); cur += 50
    torch.cuda.synchronize()
    d = (sim.field("stats_sum") - s0).to(torch.int64).sum(dim=0).cpu().numpy()
    steps, dual, kkt, tie, stall, cost, maxit, primal, fallback, big, noslip, free = (int(d[k]) for k in (0, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14))
    print(name, dict(steps=steps, contact_space=dual, kkt=kkt, tie=tie, stall=stall, cost=cost, maxit=maxit, primal=primal, fallback=fallback, big=big, free=free))
    assert steps == 2000 * n and bool(torch.isfinite(sim.field("qpos")).all())
    assert dual + primal + free == steps and kkt + tie + stall + cost + maxit == dual
    assert kkt >= 0.999 * dual and maxit == 0 and noslip == 0
    assert fallback <= primal and fallback <= 1e-3 * steps
    if name == "config2":
        assert primal <= 1e-4 * steps          # sixteen contacts: a walking fly on flat ground never leaves the contact-space solve


def test_batch_info_names_what_runs(torch_mod, monkeypatch):
    """``nmf_batch_info``: kernel family, contact-space flavour, residency, chunk plan and solver options as the batch runs
    them; explicit create options show up in it, stray ``NMF_*`` environment variables do not change it (the library reads
    the environment only under ``NMF_ALLOW_ENV=1``)."""
    from flygym_amd import HIPSimulation, make_model

    fly, world, _ = make_model()
    monkeypatch.setenv("NMF_SOLVER", "primal"); monkeypatch.setenv("NMF_SCHED", "plain")
    info = HIPSimulation(world, n_worlds=8, device=0).batch_info()
    assert info["kernel_family"] == 0 and info["contact_space_flavour"] == 1 and info["contact_space_max_contacts"] == 16
    assert info["flies_per_cu"] == 8 and info["resident_workgroups"] == 8 * 256 and info["chunked"] == 1 and info["solver_option_bits"] == 0
    assert info["kernel_vgprs"] <= 256 and 0 < info["kernel_lds_bytes"] <= 20480
    info = HIPSimulation(world, n_worlds=8, device=0, _options=dict(solver="primal", sched="plain", max_chunks=5)).batch_info()
    assert info["solver_option_bits"] == 1 and info["contact_space_flavour"] == 0 and info["chunked"] == 0 and info["max_chunks"] == 5
    bio = make_model(joints_preset="all_biological")[1]
    info = HIPSimulation(bio, n_worlds=8, device=0).batch_info()
    assert info["kernel_family"] == 4 and info["contact_space_flavour"] == 2 and info["contact_space_max_contacts"] == 13 and info["flies_per_cu"] == 8


def test_round5_entry_points_refuse_bad_arguments(torch_mod):
    """The C ABI of the round-5 entry points fails loudly (non-zero return + ``nmf_last_error``) and leaves the batch as it
    was: zero or negative step counts, an observation interval of zero, a null ring, a ring row too short for the block it
    is asked to hold, more joints / actuators than the model has, a replay table without ids, an options struct of a size this
    library does not know, a null info array."""
    torch = torch_mod
    import ctypes
    from flygym_amd import HIPSimulation, _native, make_model

    fly, world, _ = make_model()
    sim = HIPSimulation(world, n_worlds=4, device=0)
    lib, h = sim._lib, sim._batch_h
    sim.step(3)
    before = {k: sim.field(k).clone() for k in STATE}
    ring = torch.zeros((2, 4, 270), device=sim.device)
    ok_args = dict(table=None, table_steps=0, n_act_table=0, ids=None, start=0, n_steps=2, obs_every=1, n_joint=66, n_act=42, ring=ring.data_ptr(), stride=270)

    def record(**kw):
        a = dict(ok_args, **kw)
        return lib.nmf_step_record(h, a["table"], a["table_steps"], a["n_act_table"], a["ids"], a["start"], a["n_steps"], a["obs_every"],
                                   a["n_joint"], a["n_act"], a["ring"], a["stride"], None)

    bad = [dict(n_steps=0), dict(n_steps=-4), dict(obs_every=0), dict(n_steps=3, obs_every=2),      # (round 6: a trailing partial window is refused)
            dict(ring=None), dict(stride=269), dict(n_joint=67), dict(n_act=49), dict(n_joint=-1),
           dict(table=ring.data_ptr(), table_steps=10, n_act_table=42, ids=None), dict(table=ring.data_ptr(), table_steps=0, n_act_table=42, ids=ring.data_ptr())]
    for kw in bad:
        assert record(**kw) != 0, kw
        assert b"nmf_step_record" in lib.nmf_last_error(), kw
    assert lib.nmf_step(h, 0, None) != 0 and lib.nmf_step(h, -1, None) != 0
    assert lib.nmf_batch_info(h, None) != 0 and lib.nmf_batch_info(None, (ctypes.c_int32 * 16)()) != 0
    torch.cuda.synchronize()
    for k in STATE:
        assert torch.equal(sim.field(k), before[k]), k             # nothing ran
    opts = _native.BatchOptions.make()
    opts.struct_size = ctypes.sizeof(opts) + 8
    assert not lib.nmf_batch_create_ex(sim._model_h, 4, 0, ctypes.byref(opts)) and b"struct_size" in lib.nmf_last_error()
    opts.struct_size = 2
    assert not lib.nmf_batch_create_ex(sim._model_h, 4, 0, ctypes.byref(opts))
    assert not lib.nmf_batch_create_ex(sim._model_h, 0, 0, None)
    assert record() == 0                                            # and the batch still steps
    torch.cuda.synchronize()
    assert bool(torch.isfinite(ring).all()) and float(ring.abs().sum()) > 0


@pytest.mark.parametrize("kind", ["all_biological", "all_possible", "custom_tree", "tethered", "legs_only_on_blocks"])
def test_cpu_flavour_runs_noslip_on_every_kernel_family(torch_mod, oracle_lib, kind):
    """``flygym_amd.Simulation`` keeps ``option/noslip_iterations = 5`` (reference mujoco_globals.yaml:15) for EVERY model the
    reference's CPU class can step — rounds 3-4 had the pass in the contact-space solve only and stripped it elsewhere with the
    GPU class's warning.  Hybrid (ALL_BIOLOGICAL: contact-space pass while the contacts are on the legs, primal pass when head /
    abdomen touch; ALL_POSSIBLE likewise, its leg factors in HBM), a custom skeleton (primal loop + ``noslip_primal``), a tethered fly (six weld rows in A)
    and a terrain world: no warning, every step with contacts takes the pass, and sampled steps match the oracle running the
    same pass from the same state — which differs from the oracle without it by far more."""
    torch = torch_mod
    import warnings
    import flygym_amd.compose as C
    from flygym_amd import Simulation, make_model, anatomy as A
    from flygym_amd.controllers import TripodCPG
    from flygym_amd.utils.math import Rotation3D

    def build(noslip):
        upright = Rotation3D("quat", (1, 0, 0, 0))
        if kind in ("all_biological", "all_possible"):
            fly, world, _ = make_model(joints_preset=kind)
        elif kind == "tethered":
            fly = make_model()[0]; world = C.TetheredWorld(); world.add_fly(fly, (0, 0, 1.5), upright)
        elif kind == "legs_only_on_blocks":
            fly = make_model()[0]; world = C.BlocksTerrainWorld(); world.add_fly(fly, (0.3, 0.2, 0.8), upright)
        else:
            fly = C.Fly(name="t")
            bio = A.Skeleton(axis_order=A.AxisOrder.YAW_PITCH_ROLL, joint_preset=A.JointPreset.ALL_BIOLOGICAL)
            keep = [j for j in bio.anatomical_joints if not any(k in j.child.name for k in ("wing", "haltere", "abdomen"))]
            fly.add_joints(A.Skeleton(axis_order=A.AxisOrder.YAW_PITCH_ROLL, anatomical_joints=keep), neutral_pose=C.KinematicPosePreset.NEUTRAL)
            legs = A.Skeleton(axis_order=A.AxisOrder.YAW_PITCH_ROLL, joint_preset=A.JointPreset.LEGS_ONLY)
            fly.add_actuators(legs.get_actuated_dofs_from_preset("legs_active_only"), C.ActuatorType.POSITION, kp=50.0, neutral_input=C.KinematicPosePreset.NEUTRAL)
            fly.add_leg_adhesion()
            world = C.FlatGroundWorld(); world.add_fly(fly, (0, 0, 0.8), upright)
        if not noslip:
            world.noslip_iterations = 0; world._compiled = None
        return fly, world

    fly, world = build(True)
    assert world.noslip_iterations == 5
    with warnings.catch_warnings():
        warnings.simplefilter("error")                     # no "does not run noslip iterations" warning on any model
        sim = Simulation(world)
    batch = sim.batch
    assert int(batch.model["opt_solver"][1]) == 5 and batch.batch_info()["noslip_iterations"] == 5
    batch.set_leg_adhesion_states(fly.name, np.ones((1, 6), dtype=np.float32))
    table = TripodCPG(fly.get_actuated_jointdofs_order("position"), 1e-4).targets(1, 2500, device=batch.device)
    ids = batch.replay_ids(fly.name)
    batch.warmup(); batch.step_replay(table, ids, 0, 400)
    keys = ("qpos", "qvel", "ctrl", "qacc_warmstart")
    blob5, blob0 = batch.model.to_blob(), build(False)[1].compile_model().to_blob()
    worst, changed, cur, compared, primal_steps = 0.0, 0.0, 400, 0, 0
    for k in range(8):
        batch.step_replay(table, ids, cur, 19); cur += 19
        state = {kk: batch.field(kk)[0].cpu().numpy().astype(np.float64) for kk in keys}
        batch.step_replay(table, ids, cur, 1); cur += 1
        torch.cuda.synchronize()
        qacc = batch.field("qacc")[0].cpu().numpy().astype(np.float64)
        stats = batch.field("stats")[0].cpu().numpy()
        refs = {}
        for name, blob in (("noslip", blob5), ("plain", blob0)):
            r = oracle_lib.Oracle(blob, "f64", cpu_flavour=True)
            for kk in keys: r.arr(kk)[:] = state[kk]
            r.step_replay(table[0].cpu().numpy(), ids.cpu().numpy(), cur - 1, 1)
            refs[name] = r
        if refs["noslip"].ints()["ncon"] != int(stats[0]) or int(stats[0]) == 0: continue
        compared += 1
        primal_steps += (int(stats[4]) >> 6) & 1
        scale = np.abs(refs["noslip"].arr("qacc")).max()
        worst = max(worst, np.abs(qacc - refs["noslip"].arr("qacc")).max() / scale)
        changed = max(changed, np.abs(refs["noslip"].arr("qacc") - refs["plain"].arr("qacc")).max() / scale)
    ex = batch.get_solver_exits()
    print(kind, "compared", compared, "primal-loop steps among them", primal_steps, "worst", f"{worst:.2e}", "the pass moves qacc by", f"{changed:.2e}", ex)
    if kind == "tethered":                                  # no ground: the pass has nothing to do (and must not disturb the weld)
        assert ex["noslip_skipped"] == 0 and bool(torch.isfinite(batch.field("qpos")).all())
        return
    assert compared >= 5 and worst < 2e-3 and changed > 5 * worst, (compared, worst, changed)
    assert ex["noslip_skipped"] == 0
    if kind == "custom_tree":
        assert primal_steps == compared and ex["contact_space"] == 0      # the general-tree kernels have the primal loop only


def test_noslip_removes_the_creep_on_the_kernel(torch_mod, oracle_lib):
    """The closed-form anchor of the pass (tests/test_oracle_closed_form.py::test_noslip_removes_the_creep: a body on an incline
    below half the friction slope comes to REST instead of creeping at the soft rows' steady velocity) on the kernel's
    general-tree path, where the pass is ``noslip_primal``: velocity below 2 % of the creep prediction, held by a friction force
    of m g sin(theta), the same numbers as the float32 oracle."""
    torch = torch_mod
    from flygym_amd import HIPSimulation
    from tiny_models import TinyWorld, sphere_on_plane

    MASS, RADIUS, MU, SOLREF, SOLIMP, MARGIN, G = 1e-3, 0.1, 1.0, (2e-4, 1.0), (0.98, 0.99, 0.5, 0.9999, 2.0), 1e-3, 9810.0
    theta = np.arctan(0.3 * MU)
    n = np.array([np.sin(theta), 0.0, np.cos(theta)])
    m = sphere_on_plane(MASS, RADIUS, normal=n, mu=MU, solref=SOLREF, solimp=SOLIMP, margin=MARGIN, start_height=RADIUS + MARGIN, noslip_iterations=5)
    world = TinyWorld(m); world.noslip_iterations = 5
    sim = HIPSimulation(world, n_worlds=1, device=0, _cpu_flavour=True)
    plain = HIPSimulation(TinyWorld(sphere_on_plane(MASS, RADIUS, normal=n, mu=MU, solref=SOLREF, solimp=SOLIMP, margin=MARGIN, start_height=RADIUS + MARGIN)), n_worlds=1, device=0)
    sim.step(600); plain.step(600)
    torch.cuda.synchronize()
    o = oracle_lib.Oracle(m.to_blob(), "f32", cpu_flavour=True)
    o.step(600)
    v, v_plain = sim.field("qvel")[0, :3].cpu().numpy(), plain.field("qvel")[0, :3].cpu().numpy()
    creep = np.abs(v_plain).max()
    assert creep > 1e-3                                              # without the pass the body creeps down the slope
    assert np.abs(v).max() < 0.02 * creep                            # with it: at rest
    assert np.abs(sim.field("qacc")[0, :3].cpu().numpy()).max() < 1e-3 * G
    np.testing.assert_allclose(sim.field("qpos")[0, :3].cpu().numpy(), np.asarray(o.qpos[:3], dtype=np.float64), atol=2e-5)
    assert sim.get_solver_exits()["noslip_skipped"] == 0 and sim.get_solver_exits()["primal_loop"] >= 590
