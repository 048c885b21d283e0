"""Round 4: the contact-space constraint solve (csrc/nmf_dual.h) against the primal Newton loop and the oracle, and the
collision fix it exposed.  GPU tests, through the C ABI."""
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


def _walkers(n, solver, torch, preset="legs_only"):
    from flygym_amd import HIPSimulation, make_model
    from flygym_amd.controllers import TripodCPG

    fly, world, _ = make_model(joints_preset=preset)
    sim = HIPSimulation(world, n_worlds=n, device=0, _options=dict(solver=solver))          # an explicit create option (nmf_batch_create_ex)
    table = TripodCPG(fly.get_actuated_jointdofs_order("position"), 1e-4).targets(n, 2500, device=sim.device)
    ids = sim.replay_ids(fly.name)
    sim.set_leg_adhesion_states(fly.name, np.ones((n, 6), dtype=np.float32))
    return sim, table, ids


@pytest.mark.parametrize("preset,n", [("legs_only", 2048), ("all_biological", 1024)])
def test_contact_space_solve_reaches_the_primal_loops_optimum(torch_mod, oracle_lib, preset, n):
    """The same walking flies on three builds of the solver — contact-space with the active-set history (default),
    contact-space from the start point's own sign pattern (NMF_SOLVER=nohist: MuJoCo's Newton iterates, row by row) and the
    primal loop (NMF_SOLVER=primal) — stepped from IDENTICAL states: one optimum, so the accelerations agree to float32
    accuracy, the contact-space ones no further from the float64 oracle than the primal ones; the history only shortens the
    way (fewer eliminations than Newton iterations), it never changes where it ends.  ALL_BIOLOGICAL runs the hybrid
    kernels' flavour (no warm-start term, the history as a list, steps with more than 10 contacts on the primal loop)."""
    torch = torch_mod
    sims = {k: _walkers(n, k, torch, preset) for k in ("", "nohist", "primal")}
    lead, table, ids = sims[""]
    lead.warmup(); lead.step_replay(table, ids, 0, 850)
    keys = ("qpos", "qvel", "ctrl", "qacc_warmstart")
    blob = lead.model.to_blob()
    it_sum = {k: 0.0 for k in sims}
    worst = {k: 0.0 for k in sims}
    worst_between = {"": 0.0, "nohist": 0.0}
    compared = skipped = 0          # oracle comparisons made / not made because the oracle's contact count differs (a near-tie at the margin)
    cur = 850
    for checkpoint in range(6):
        lead.step_replay(table, ids, cur, 37); cur += 37
        state = {k: lead.field(k).clone() for k in keys}
        qacc = {}
        for name, (sim, _, _) in sims.items():
            if sim is not lead:
                for k in keys: sim.field(k)[:] = state[k]
            if name != "": sim.step_replay(table, ids, cur, 1)
        lead.step_replay(table, ids, cur, 1); cur += 1
        torch.cuda.synchronize()
        for name, (sim, _, _) in sims.items():
            qacc[name] = sim.field("qacc").cpu().numpy().astype(np.float64)
            it_sum[name] += float(sim.field("stats")[:, 1].double().mean().item())
        nc = {name: sim.field("stats")[:, 0].cpu().numpy() for name, (sim, _, _) in sims.items()}
        assert np.array_equal(nc[""], nc["primal"]) and np.array_equal(nc[""], nc["nohist"])          # same collision stage, same state
        scale = np.abs(qacc["primal"]).max(axis=1)
        for name in ("", "nohist"):
            dev = np.abs(qacc[name] - qacc["primal"]).max(axis=1) / scale
            worst_between[name] = max(worst_between[name], float(dev.max()))
            # the stated tolerance (the primal loop is the less accurate of the two); worst seen over 6 x 2048 states: 4e-3
            assert np.median(dev) < 2e-4 and np.quantile(dev, 0.99) < 2e-3 and dev.max() < 1e-2, (name, np.sort(dev)[-4:])
        for w in np.random.default_rng(checkpoint).choice(n, size=12, replace=False):
            r = oracle_lib.Oracle(blob, "f64")
            for k in keys: r.arr(k)[:] = state[k][w].cpu().numpy().astype(np.float64)
            r.step_replay(table[w].cpu().numpy(), ids.cpu().numpy(), cur - 1, 1)
            if r.ints()["ncon"] != int(nc[""][w]): skipped += 1; continue
            compared += 1
            a = r.arr("qacc")
            for name in sims:
                worst[name] = max(worst[name], float(np.abs(qacc[name][w] - a).max() / np.abs(a).max()))
    its = {k: v / 6 for k, v in it_sum.items()}
    print("iterations per step", {k or "default": round(v, 2) for k, v in its.items()}, "worst |qacc - oracle| / max", {k or "default": f"{v:.1e}" for k, v in worst.items()},
          "worst between solvers", {k or "default": f"{v:.1e}" for k, v in worst_between.items()}, "oracle comparisons", compared, "skipped", skipped)
    assert skipped <= 2 and compared >= 70                         # 72 sampled states: nearly all of them are compared
    assert max(worst.values()) < 2e-3
    assert worst[""] < 2.0 * worst["primal"] + 1e-4                 # the contact-space solve is at least as accurate
    if preset == "legs_only":
        assert its[""] < 0.75 * its["nohist"] and abs(its["nohist"] - its["primal"]) < 0.3      # same Newton iterates without the history
    else:       # (starts from the unconstrained acceleration, not from the better of it and the warm start: a little shorter still)
        assert its[""] < 0.85 * its["nohist"] and its["nohist"] < its["primal"] + 0.05


def test_hull_vertex_on_a_cell_edge_keeps_its_scan_distance(torch_mod, oracle_lib):
    """A state from a walk over BlocksTerrainWorld where a tarsus hull's deepest vertex sits within rounding of a cell edge.
    Round 3's collision stage evaluated that vertex's distance twice; the second evaluation took the other side of the tie
    (a side face owns the vertex: top distance kFar) and stored a contact 1e30 mm away, which the primal loop carried as a
    forever-inactive row — the step's accelerations were off by a factor of two, unnoticed under the percentage bars.  The
    contact-space solve, started from the previous step's active set, turned it into a NaN and so found it.  Two steps from
    the saved state (the first one builds the history): contact list and accelerations of the second like the oracle's."""
    torch = torch_mod
    import flygym_amd.compose as C
    from flygym_amd import HIPSimulation, make_model
    from flygym_amd.utils.math import Rotation3D

    d = np.load(GOLD / "terrain_edge_tie_state.npz")
    fly = make_model(joints_preset="legs_only")[0]
    world = C.BlocksTerrainWorld()
    world.add_fly(fly, (0, 0, 0.8), Rotation3D("quat", (1, 0, 0, 0)))
    sim = HIPSimulation(world, n_worlds=4, device=0)
    ids = sim.replay_ids(fly.name)
    keys = ("qpos", "qvel", "ctrl", "qacc_warmstart")
    rows = torch.as_tensor(d["rows"], device=sim.device)[None].repeat(4, 1, 1).contiguous()
    for k in keys: sim.field(k)[:] = torch.as_tensor(d[k], device=sim.device)[None, :]
    sim.step_replay(rows, ids, 0, 1)
    state = {k: sim.field(k)[0].cpu().numpy().astype(np.float64) for k in keys}
    sim.step_replay(rows, ids, 1, 1)
    torch.cuda.synchronize()
    qacc = sim.field("qacc").cpu().numpy(); stats = sim.field("stats").cpu().numpy(); geom = sim.field("contact_geom").cpu().numpy()
    assert np.isfinite(qacc).all() and stats[0, 1] < 10
    r = oracle_lib.Oracle(sim.model.to_blob(), "f64")
    for k in keys: r.arr(k)[:] = state[k]
    r.step_replay(d["rows"], ids.cpu().numpy(), 1, 1)
    nc = int(stats[0, 0])
    assert nc == r.ints()["ncon"] == 7 and geom[0, :nc].astype(int).tolist() == r.ints()["con_geom"]
    assert min(r.arr("con_dist")) > -0.05                         # every contact a real one
    a = r.arr("qacc")
    assert np.abs(qacc[0] - a).max() < 2e-3 * np.abs(a).max()


def test_noslip_pass_of_the_cpu_flavour(torch_mod, oracle_lib):
    """``flygym_amd.Simulation`` — the drop-in for the reference's CPU class — keeps ``option/noslip_iterations = 5``
    (``mujoco_globals.yaml:15``) and runs MuJoCo's documented friction-only post-pass after the Newton solve; the batched
    class strips it, as the reference's does.  The pass itself is anchored on the oracle by a closed-form case
    (tests/test_oracle_closed_form.py::test_noslip_removes_the_creep: a one-body model, which the kernel steps on its
    general-tree path where the pass does not exist); here the benchmark fly walks on the kernel and every sampled step is
    compared with the oracle running the same pass from the same state — and with the oracle WITHOUT it: the pass moves the
    accelerations by far more than the two engines differ."""
    torch = torch_mod
    import warnings
    from flygym_amd import Simulation, make_model
    from flygym_amd.controllers import TripodCPG
    fly, world, _ = make_model()
    assert world.noslip_iterations == 5
    with warnings.catch_warnings():
        warnings.simplefilter("error")                     # the CPU flavour must not warn about noslip any more
        sim = Simulation(world)
    assert sim.batch.model["opt_solver"][1] == 5
    batch = sim.batch
    batch.set_leg_adhesion_states(fly.name, np.ones((1, 6), dtype=np.float32))
    table = TripodCPG(fly.get_actuated_jointdofs_order("position"), 1e-4).targets(1, 2500, device=batch.device)
    ids = batch.replay_ids(fly.name)
    batch.warmup(); batch.step_replay(table, ids, 0, 700)
    keys = ("qpos", "qvel", "ctrl", "qacc_warmstart")
    blob5 = batch.model.to_blob()
    world0 = make_model()[1]; world0.noslip_iterations = 0
    blob0 = world0.compile_model().to_blob()
    worst, changed, cur, skipped = 0.0, 0.0, 700, 0
    for k in range(12):
        batch.step_replay(table, ids, cur, 23); cur += 23
        state = {kk: batch.field(kk)[0].cpu().numpy().astype(np.float64) for kk in keys}
        batch.step_replay(table, ids, cur, 1); cur += 1
        torch.cuda.synchronize()
        qacc = batch.field("qacc")[0].cpu().numpy().astype(np.float64)
        refs = {}
        for name, blob in (("noslip", blob5), ("plain", blob0)):
            r = oracle_lib.Oracle(blob, "f64", cpu_flavour=True)
            for kk in keys: r.arr(kk)[:] = state[kk]
            r.step_replay(table[0].cpu().numpy(), ids.cpu().numpy(), cur - 1, 1)
            refs[name] = r
        if refs["noslip"].ints()["ncon"] != int(batch.field("stats")[0, 0].item()): skipped += 1; continue
        scale = np.abs(refs["noslip"].arr("qacc")).max()
        worst = max(worst, np.abs(qacc - refs["noslip"].arr("qacc")).max() / scale)
        changed = max(changed, np.abs(refs["noslip"].arr("qacc") - refs["plain"].arr("qacc")).max() / scale)
    assert skipped <= 1, skipped                                   # (a contact at the margin may differ: at most one of the 12 samples)
    assert worst < 2e-3 and changed > 10 * worst, (worst, changed)
    assert int(batch.field("stats_sum")[0, 3].item()) == 0 and batch.get_solver_exits()["noslip_skipped"] == 0         # every step with contacts took the pass


def test_cells_narrower_than_a_hulls_footprint(torch_mod, oracle_lib):
    """The collision stage's cull takes the highest top under a geom's footprint.  Round 3 read it from 3 x 3 samples a
    footprint radius apart, which is a bound only while no raised cell is narrower than that radius; the terrain classes
    accept any widths, and the thorax / abdomen hulls' footprints (0.7 mm) are wider than the blocks of this world
    (0.5 mm blocks, 0.3 mm gaps): all three samples of a row can land in gaps.  The walk along the cells (round 4) meets
    every cell.  A fly dropped on its side onto that terrain: states of the float64 oracle's fall in which the body's hulls
    touch, one step from each on the engine — the same contacts, the same accelerations."""
    torch = torch_mod
    import flygym_amd.compose as C
    from flygym_amd import HIPSimulation, make_model
    from flygym_amd.utils.math import Rotation3D

    fly = make_model(joints_preset="legs_only")[0]
    world = C.GappedTerrainWorld(block_width=0.5, gap_width=0.3)
    world.add_fly(fly, (0.1, 0, 0.9), Rotation3D("quat", (0.7071, 0.7071, 0, 0)))       # on its side: it comes down on thorax and head
    model = world.compile_model()
    blob = model.to_blob()
    big = np.nonzero(np.asarray(model["geom_bsphere"]).reshape(-1, 4)[:, 3] > 0.5)[0].tolist()
    assert big, "no wide hull in the contact set"
    o = oracle_lib.Oracle(blob, "f64")
    states = []
    for k in range(1500):
        o.step(1)
        g = o.ints()["con_geom"]
        if any(x in big for x in g) and (not states or k - states[-1][0] >= 20):
            states.append((k, o.qpos.copy(), o.qvel.copy(), o.ctrl.copy(), o.arr("qacc_warmstart").copy()))
        if len(states) == 8:
            break
    assert len(states) >= 3, "the body never reached the ground"
    n = len(states)
    sim = HIPSimulation(world, n_worlds=n, device=0)
    for name, idx in (("qpos", 1), ("qvel", 2), ("ctrl", 3), ("qacc_warmstart", 4)):
        sim.field(name)[:] = torch.as_tensor(np.stack([s[idx] for s in states]), dtype=torch.float32, device=sim.device)
    sim.step(1)
    torch.cuda.synchronize()
    qacc, stats, geom = sim.field("qacc").cpu().numpy(), sim.field("stats").cpu().numpy(), sim.field("contact_geom").cpu().numpy()
    wide = 0
    for w, st in enumerate(states):
        r = oracle_lib.Oracle(blob, "f64")
        r.qpos[:] = st[1]; r.qvel[:] = st[2]; r.ctrl[:] = st[3]; r.arr("qacc_warmstart")[:] = st[4]
        r.step(1)
        nc = int(stats[w, 0])
        assert nc == r.ints()["ncon"], f"state {w}: engine {geom[w, :nc].astype(int).tolist()}, oracle {r.ints()['con_geom']}"
        assert geom[w, :nc].astype(int).tolist() == r.ints()["con_geom"]
        scale = np.abs(r.arr("qacc")).max()
        assert np.abs(qacc[w] - r.arr("qacc")).max() < 5e-3 * scale, f"state {w}"
        wide += sum(1 for x in r.ints()["con_geom"] if x in big)
    assert wide >= 2


def test_max_contacts_is_the_contact_capacity(torch_mod, oracle_lib):
    """``HIPSimulation(max_contacts=...)`` sizes the contact list as the reference's argument sizes MJWarp's
    (``warp/simulation.py:50-56``), up to the engine's 48.  A standing fly makes 6-12 contacts: with a capacity of 4 the
    engine keeps the four with the lowest geom indices, solves with those — like the oracle given the same capacity — and
    counts the step as overflowed; the default capacity keeps them all.  ``strict_contacts`` refuses a model whose contact
    set could exceed the capacity."""
    torch = torch_mod
    from flygym_amd import HIPSimulation, make_model

    fly, world, _ = make_model()
    full = HIPSimulation(world, n_worlds=2, device=0)
    assert full.contact_capacity == 48 and full.contact_bound > 48 and full.max_contacts == 500
    full.step(400)
    state = {k: full.field(k)[0].cpu().numpy().astype(np.float64) for k in ("qpos", "qvel", "ctrl", "qacc_warmstart")}
    n_full = int(full.get_solver_stats()[0, 0].item())
    assert n_full >= 6 and full.overflow_steps() == 0
    sim = HIPSimulation(make_model()[1], n_worlds=2, max_contacts=4, device=0)
    assert sim.contact_capacity == 4
    for k, v in state.items(): sim.field(k)[:] = torch.as_tensor(v, dtype=torch.float32, device=sim.device)[None]
    sim.step(1)
    torch.cuda.synchronize()
    stats = sim.get_solver_stats().cpu().numpy()
    assert stats[0, 0] == 4 and stats[0, 2] == 1 and sim.overflow_steps() == 2
    r = oracle_lib.Oracle(sim.model.to_blob(), "f64")
    r.set_max_contacts(4)
    for k, v in state.items(): r.arr(k)[:] = v
    r.step(1)
    assert r.ints()["ncon"] == 4 and r.ints()["overflow"] == 1
    assert sim.field("contact_geom")[0, :4].cpu().numpy().astype(int).tolist() == r.ints()["con_geom"]
    a = r.arr("qacc")
    assert np.abs(sim.field("qacc")[0].cpu().numpy() - a).max() < 2e-3 * np.abs(a).max()
    with pytest.raises(ValueError, match="contacts in one step"):
        HIPSimulation(make_model()[1], n_worlds=2, device=0, strict_contacts=True)
    with pytest.raises(ValueError):
        HIPSimulation(make_model()[1], n_worlds=2, device=0, max_contacts=0)


@pytest.mark.parametrize("state", ["solver_tie_state_0.npz", "solver_tie_state_1.npz"])
def test_a_tie_row_does_not_send_the_solve_into_the_noise(torch_mod, oracle_lib, state):
    """Two states from a 20 000-step soak of config 5 (mixed terrain, 20 x gait adhesion, 10 and 12 contacts) on which the
    contact-space solve of this round's first versions blew up — once in 40 M steps, found by ``scripts/gpu_soak.py``, not by
    any parity test: a row whose residual is zero to rounding flipped in and out of the active set, the loop went on past the
    optimum, and two eliminations later the line search divided a slope by a curvature of 4e-6 that was the difference of two
    sums of 4e-5 — a step of 4e8, accelerations of 3e9, a fly leaving the scene.  The loop now recognises the pivot set of two
    eliminations ago (a tie) and takes that target, and no step goes further than four times the way to its target.  From the
    saved state: the oracle's accelerations, on every solver variant."""
    torch = torch_mod
    import flygym_amd.compose as C
    from flygym_amd import HIPSimulation, make_model
    from flygym_amd.utils.math import Rotation3D

    d = np.load(GOLD / state)
    keys = ("qpos", "qvel", "ctrl", "qacc_warmstart")
    ref = None
    for solver in ("", "nohist", "primal"):
        fly = make_model()[0]
        world = C.MixedTerrainWorld()
        world.add_fly(fly, (0, 0, 0.8), Rotation3D("quat", (1, 0, 0, 0)))
        sim = HIPSimulation(world, n_worlds=2, device=0, _options=dict(solver=solver))
        ids = sim.replay_ids(fly.name, with_adhesion=True)
        rows = torch.as_tensor(d["rows"], device=sim.device)[None].repeat(2, 1, 1).contiguous()
        for k in keys: sim.field(k)[:] = torch.as_tensor(d[k], device=sim.device)[None, :]
        sim.step_replay(rows, ids, int(d["cur"]), 1)
        torch.cuda.synchronize()
        if ref is None:
            ref = oracle_lib.Oracle(sim.model.to_blob(), "f64")
            for k in keys: ref.arr(k)[:] = d[k].astype(np.float64)
            ref.step_replay(d["rows"], ids.cpu().numpy(), int(d["cur"]), 1)
        a = ref.arr("qacc")
        qacc, stats = sim.field("qacc")[0].cpu().numpy(), sim.field("stats")[0].cpu().numpy()
        assert int(stats[0]) == ref.ints()["ncon"] and stats[1] <= 8
        assert np.abs(qacc - a).max() < 2e-3 * np.abs(a).max(), solver


def test_cpu_flavour_runs_the_noslip_pass_on_every_step(torch_mod, oracle_lib):
    """7 % of a walking fly's steps have 13-15 contacts — more than rounds 3-4's contact-space solve took at eight flies per CU (12:
    the triangle of A by rows).  Since round 5 the solve keeps the Gram matrix of the contact directions and takes 16 on every
    kernel, so the CPU flavour's noslip pass, which lives in that solve, runs on EVERY step.  256 walkers:
    no step without the pass over 3000 steps although the contact count passes 12 on the way; steps with 13 contacts and more
    against the oracle running the same pass."""
    torch = torch_mod
    from flygym_amd import HIPSimulation, make_model
    from flygym_amd.controllers import TripodCPG

    n = 256
    fly, world, _ = make_model()
    sim = HIPSimulation(world, n_worlds=n, device=0, _cpu_flavour=True)
    assert world.noslip_iterations == 5
    table = TripodCPG(fly.get_actuated_jointdofs_order("position"), 1e-4).targets(n, 2500, device=sim.device)
    ids = sim.replay_ids(fly.name)
    sim.set_leg_adhesion_states(fly.name, np.ones((n, 6), dtype=np.float32))
    sim.warmup(); sim.step_replay(table, ids, 0, 850)
    keys = ("qpos", "qvel", "ctrl", "qacc_warmstart")
    blob = sim.model.to_blob()
    cur, many, worst, compared, skipped = 850, 0, 0.0, 0, 0
    for k in range(40):
        sim.step_replay(table, ids, cur, 49); cur += 49
        state = {kk: sim.field(kk).clone() for kk in keys}
        sim.step_replay(table, ids, cur, 1); cur += 1
        torch.cuda.synchronize()
        stats = sim.field("stats").cpu().numpy()
        big = np.nonzero(stats[:, 0] >= 13)[0]
        many += len(big)
        for w in big[:2]:
            r = oracle_lib.Oracle(blob, "f64", cpu_flavour=True)
            for kk in keys: r.arr(kk)[:] = state[kk][w].cpu().numpy().astype(np.float64)
            r.step_replay(table[w].cpu().numpy(), ids.cpu().numpy(), cur - 1, 1)
            if r.ints()["ncon"] != int(stats[w, 0]): skipped += 1; continue
            compared += 1
            a = r.arr("qacc")
            worst = max(worst, float(np.abs(sim.field("qacc")[w].cpu().numpy() - a).max() / np.abs(a).max()))
    ss = sim.field("stats_sum").cpu().numpy()
    assert many >= 20, many                                           # the walk does pass 12 contacts
    assert compared >= 20 and skipped <= max(2, compared // 10), (compared, skipped)
    assert int(ss[:, 3].sum()) == 0 and int(ss[:, 13].sum()) == 0 and int(ss[:, 0].min()) == cur + 500      # ... and no step went without the pass
    assert 0.0 < worst < 2e-3, worst
