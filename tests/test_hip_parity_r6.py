"""Round 6 parity tests on MI355X (through the C ABI): BASELINE config 1 on the engine the CPU class really runs (noslip on,
a fresh world), one warm-start rule for the CPU flavour on both solver paths, the reference benchmark's all-capsule series at
full size, and the contact list beyond one wave of contacts.

Bars are those of rounds 3-5 or tighter (round-5 verdict item 1c); figures against the tight bars go to the parity ledger
(``tests/ledger.py``)."""

import numpy as np
import pytest

from ledger import report

pytestmark = pytest.mark.gpu

KEYS = ("qpos", "qvel", "ctrl", "qacc_warmstart")


@pytest.fixture(scope="module")
def torch_mod():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    return torch


def _push(sim, torch, state):
    for k, v in zip(KEYS, state):
        sim.field(k)[:] = torch.as_tensor(np.asarray(v), dtype=torch.float32, device=sim.device)


def test_config1_rollout_runs_the_cpu_class(torch_mod, oracle_lib):
    """BASELINE config 1 as the reference's CPU class runs it (``src/flygym/simulation.py:74-76`` under ``mujoco_globals.yaml:15``:
    Newton + 5 noslip sweeps): one fly, flat ground, adhesion on, the reference's warm-up (500 steps) and then >= 700 steps of the
    Spotlight replay — ``flygym_amd.Simulation`` on a world no other test has touched, against ``Oracle(cpu_flavour=True)``.

    Three comparisons, all with the pass on in both engines: (1) the first 100 driven steps through the reference's own loop
    (``set_actuator_inputs`` + ``step()`` per step) free-running from reset; (2) the whole 1200-step rollout in re-synchronised
    20-step segments (contact-rich rollouts are chaotic: a contact crossing its margin one step apart kicks the stiff contact
    spring differently): every segment must end within float32 rounding of the float64 oracle started from the same state, or be
    one where the float32 ORACLE itself leaves the float64 one; (3) the free-running rollout's distance from the oracle's at
    steps 100 .. 700, reported (ledger) and bounded where it is not chaotic."""
    torch = torch_mod
    import warnings
    from flygym_amd import Simulation, make_model
    from flygym_amd.replay import ReplayTargetData

    fly, world, _ = make_model()
    assert world.noslip_iterations == 5
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        sim = Simulation(world, device=0)
    assert world.noslip_iterations == 5 and sim.batch.model["opt_solver"][1] == 5 and sim.batch.batch_info()["noslip_iterations"] == 5
    order = fly.get_actuated_jointdofs_order("position")
    n_drive = 700
    targets = ReplayTargetData(sim.timestep, order).make_target_angles_all_worlds(1, n_drive)[0]
    blob = sim.batch.model.to_blob()
    o = oracle_lib.Oracle(blob, "f64", cpu_flavour=True)
    sim.set_leg_adhesion_states(fly.name, np.ones(6))
    o.ctrl[42:] = 1.0
    sim.warmup(); o.step(500)
    assert abs(sim.time - 0.05) < 1e-6
    after_warmup = float(np.abs(sim.batch.field("qpos")[0].cpu().numpy() - o.qpos).max())
    assert after_warmup < 5e-5, after_warmup
    ids = np.arange(42, dtype=np.int32)
    # (1) the reference's loop, free-running
    for k in range(100):
        sim.set_actuator_inputs(fly.name, "position", targets[k])
        sim.step()
    o.step_replay(targets, ids, 0, 100)
    free = {100: float(np.abs(sim.batch.field("qpos")[0].cpu().numpy() - o.qpos).max())}
    assert free[100] < 5e-4, free
    # (3) ... and on, in fused launches
    table = torch.as_tensor(targets[None], device=sim.batch.device).contiguous()
    dev_ids = sim.batch.replay_ids(fly.name)
    for upto in range(200, n_drive + 1, 100):
        sim.batch.step_replay(table, dev_ids, upto - 100, 100)
        o.step_replay(targets, ids, upto - 100, 100)
        free[upto] = float(np.abs(sim.batch.field("qpos")[0].cpu().numpy() - o.qpos).max())
    assert sim.batch.get_solver_exits()["noslip_skipped"] == 0 and sim.batch.overflow_steps() == 0
    assert np.isfinite(list(free.values())).all()
    # (2) re-synchronised segments over warm-up + replay
    base = oracle_lib.Oracle(blob, "f64", cpu_flavour=True)
    base.ctrl[42:] = 1.0
    o32 = oracle_lib.Oracle(blob, "f32", cpu_flavour=True)
    seg, tight, by_oracle, worst, off = 20, 0, 0, 0.0, []
    n_seg = (500 + n_drive) // seg
    for i in range(n_seg):
        state = [base.arr(k).copy() for k in KEYS]
        _push(sim.batch, torch, [s[None] for s in state])
        for k, v in zip(KEYS, state): o32.arr(k)[:] = v
        start = i * seg - 500
        if start < 0:
            sim.batch.step(seg); base.step(seg); o32.step(seg)
        else:
            sim.batch.step_replay(table, dev_ids, start, seg); base.step_replay(targets, ids, start, seg); o32.step_replay(targets, ids, start, seg)
        d = float(np.abs(sim.batch.field("qpos")[0].cpu().numpy() - base.qpos).max())
        d32 = float(np.abs(o32.qpos - base.qpos).max())
        worst = max(worst, d)
        if d < 2e-5: tight += 1
        elif d < 5.0 * d32 + 2e-5: by_oracle += 1
        else: off.append((i, d, d32))
    report("config1_cpu_flavour_rollout", after_warmup=after_warmup, free_running=free, segments=n_seg, segments_tight=tight,
           segments_where_the_f32_oracle_leaves_too=by_oracle, unexplained=len(off), worst_segment=worst)
    assert not off, off
    # (measured, round 6: all 60 segments within 8.3e-7, the free-running rollout within 3.2e-6 of the oracle's through all 700 driven
    # steps — this input is not chaotic over the horizon, so the free run is held to float32 rounding too)
    assert tight >= n_seg - 2, (tight, n_seg)
    assert max(free.values()) < 5e-5, free
    # the reference's own invariants on this object (tests/core/test_simulation.py)
    active, force, torque, pos, normal, tangent = sim.get_ground_contact_info(fly.name)
    assert active.shape == (6,) and force.shape == (6, 3) and tangent.shape == (6, 3)
    assert active.sum() >= 3 and force[:, 2].sum() > 0


@pytest.mark.parametrize("solver", ["", "primal"])
def test_cpu_flavour_warm_start_is_the_main_solvers_result(torch_mod, oracle_lib, solver):
    """One rule on both solver paths (round-5 verdict 1b, advisor): with the noslip pass on, ``qacc_warmstart`` — caller-visible
    state (``NMF_QACC_WARMSTART``) — is the MAIN solver's acceleration, saved before the pass (MuJoCo's ``mj_fwdConstraint`` order;
    ``oracle/nmf_oracle.c::step``), and ``qacc`` is the acceleration with the pass's forces.  Walking states of the CPU flavour,
    one step on the contact-space path (default) and on the primal loop: both fields against the oracle's, and the two differ by
    far more than engine and oracle do."""
    torch = torch_mod
    from flygym_amd import HIPSimulation, make_model
    from flygym_amd.controllers import TripodCPG

    n = 64
    fly, world, _ = make_model()
    sim = HIPSimulation(world, n_worlds=n, device=0, _cpu_flavour=True, _options=dict(solver=solver))
    assert world.noslip_iterations == 5 and sim.batch_info()["noslip_iterations"] == 5
    table = TripodCPG(fly.get_actuated_jointdofs_order("position"), 1e-4).targets(n, 2500, device=sim.device)
    ids = sim.replay_ids(fly.name)
    sim.set_leg_adhesion_states(fly.name, np.ones((n, 6), dtype=np.float32))
    sim.warmup(); sim.step_replay(table, ids, 0, 900)
    blob = sim.model.to_blob()
    cur, worst_ws, worst_acc, moved, compared = 900, 0.0, 0.0, [], 0
    for k in range(3):
        sim.step_replay(table, ids, cur, 37); cur += 37
        state = {kk: sim.field(kk).cpu().numpy().astype(np.float64) for kk in KEYS}
        sim.step_replay(table, ids, cur, 1); cur += 1
        torch.cuda.synchronize()
        ws, qacc, stats = sim.field("qacc_warmstart").cpu().numpy(), sim.field("qacc").cpu().numpy(), sim.field("stats").cpu().numpy()
        for w in range(0, n, 4):
            r = oracle_lib.Oracle(blob, "f64", cpu_flavour=True)
            for kk in KEYS: r.arr(kk)[:] = state[kk][w]
            r.step_replay(table[w].cpu().numpy(), ids.cpu().numpy(), cur - 1, 1)
            if r.ints()["ncon"] != int(stats[w, 0]) or r.ints()["ncon"] == 0: continue
            compared += 1
            a, a_ws = r.arr("qacc"), r.arr("qacc_warmstart")
            scale = np.abs(a).max()
            worst_acc = max(worst_acc, float(np.abs(qacc[w] - a).max() / scale))
            worst_ws = max(worst_ws, float(np.abs(ws[w] - a_ws).max() / scale))
            moved.append(float(np.abs(a - a_ws).max() / scale))
    exits = sim.get_solver_exits()
    report("cpu_flavour_warm_start", solver=solver or "contact space", compared=compared, worst_qacc=worst_acc, worst_warm_start=worst_ws,
           pass_moves_median=float(np.median(moved)), primal_steps=exits["primal_loop"], contact_space_steps=exits["contact_space"])
    assert compared >= 40
    assert (exits["contact_space"] == 0) == (solver == "primal")
    assert worst_acc < 2e-3 and worst_ws < 2e-3, (worst_acc, worst_ws)
    assert np.median(moved) > 10 * max(worst_ws, worst_acc)           # the two fields are different things


def test_all_capsule_series_at_full_size(torch_mod, oracle_lib):
    """The reference benchmark's second series (``scripts/dev/run_gpu_benchmark.py:15-24`` sweeps ``simplify_geom in [False, True]``,
    ``time_gpu_simulation.py:29-35``: every collision geom a capsule) at BASELINE config 2's size: 4096 flies under the reference's
    replay protocol; at three checkpoints 24 worlds' own states go to both oracles and the next step is compared — contact lists
    bit-exact, accelerations to float32 accuracy (the bars of the mesh-hull test, ``test_hip_parity_r3.py``)."""
    torch = torch_mod
    from flygym_amd import HIPSimulation, make_model
    from flygym_amd.replay import ReplayTargetData

    n = 4096
    fly, world, _ = make_model(simplify_geom=True)
    sim = HIPSimulation(world, n_worlds=n, device=0)
    assert int(np.asarray(sim.model["geom_hullnum"]).max()) == 0          # no hull anywhere: capsules only
    order = fly.get_actuated_jointdofs_order("position")
    table_np = ReplayTargetData(1e-4, order).make_target_angles_all_worlds(n, 1000)
    table = torch.as_tensor(table_np, device=sim.device)
    ids = sim.replay_ids(fly.name); ids_np = ids.cpu().numpy()
    sim.set_leg_adhesion_states(fly.name, np.ones((n, 6), dtype=np.float32))
    sim.warmup()
    blob = sim.model.to_blob()
    rng = np.random.default_rng(21)
    cur, same, total, devs, beyond = 0, 0, 0, [], 0
    for checkpoint in range(3):
        sim.step_replay(table, ids, cur, 250); cur += 250
        picks = rng.choice(n, size=24, replace=False)
        sel = torch.as_tensor(picks, device=sim.device)
        before = {k: sim.field(k)[sel].cpu().numpy().astype(np.float64) for k in KEYS}
        sim.step_replay(table, ids, cur, 1); cur += 1
        torch.cuda.synchronize()
        qacc, stats, geom = sim.field("qacc").cpu().numpy(), sim.field("stats").cpu().numpy(), sim.field("contact_geom").cpu().numpy()
        for j, w in enumerate(picks):
            ref = {}
            for prec in ("f64", "f32"):
                r = oracle_lib.Oracle(blob, prec)
                for k in KEYS: r.arr(k)[:] = before[k][j]
                r.step_replay(table_np[w], ids_np, cur - 1, 1)
                ref[prec] = r
            nc = int(stats[w, 0])
            mine = geom[w, :nc].astype(int).tolist()
            total += 1
            same += any(mine == r.ints()["con_geom"] for r in ref.values())
            if mine == ref["f64"].ints()["con_geom"]:
                a = ref["f64"].arr("qacc")
                scale = max(np.abs(a).max(), 1e4)
                dev = float(np.abs(qacc[w] - a).max() / scale)
                dev32 = float(np.abs(ref["f32"].arr("qacc") - a).max() / scale) if mine == ref["f32"].ints()["con_geom"] else 0.0
                devs.append(dev)
                assert dev < max(2e-3, 2.0 * dev32), f"world {w}: {dev:.2e} (float32 oracle {dev32:.2e})"
                beyond += dev >= 2e-3
    devs = np.array(devs)
    report("all_capsule_series_4096", lists_equal=same, total=total, compared=len(devs), median=float(np.median(devs)), worst=float(devs.max()),
           beyond_2e3=int(beyond), mean_contacts=float(stats[:, 0].mean()))
    assert total == 72 and same >= total - 1 and len(devs) >= total - 2
    assert np.median(devs) < 2e-4
    assert float(stats[:, 0].mean()) > 3 and sim.overflow_steps() == 0 and bool(torch.isfinite(sim.field("qpos")).all())


def test_a_vision_tick_is_one_captured_graph(torch_mod):
    """``nmf_eye_plan_create`` / ``nmf_eye_render_planned`` (round-5 verdict weak 8, advisor): the renderer's visit plan is an explicit
    handle with its own copies of the id map, run plan and pale flags, built once and synchronously; a render is then argument
    checks + ONE kernel launch.  So a whole vision tick — 20 physics steps under the control table and both eyes of every fly
    ray-cast to ommatidia readings — captures as ONE hipGraph on first use of the renderer's render call; replayed ticks equal
    eager ticks bit for bit.  The handle-less ``nmf_eye_render`` gives the same readings, and keeps them when the caller scribbles
    over its pale buffer afterwards (the plan holds copies; a new address makes a new plan)."""
    torch = torch_mod
    import ctypes
    from flygym_amd import HIPSimulation, _native, make_model
    from flygym_amd.controllers import TripodCPG
    from flygym_amd.vision import EyeRenderer

    n = 64
    sims, eyes, outs = [], [], []
    for _ in range(2):
        fly, world, _ = make_model()
        sim = HIPSimulation(world, n_worlds=n, device=0)
        sim.set_leg_adhesion_states(fly.name, np.ones((n, 6), dtype=np.float32))
        sim.warmup()
        sims.append(sim); eyes.append(EyeRenderer(sim, fly.name))
        outs.append(torch.zeros((n, 2, eyes[-1].retina.num_ommatidia, 2), device=sim.device))
    table = TripodCPG(fly.get_actuated_jointdofs_order("position"), 1e-4).targets(n, 2500, device=sims[0].device)
    ids = sims[0].replay_ids(fly.name)
    eager, graphed = sims
    # the graphed tick: captured BEFORE the renderer has ever rendered (no plan building inside the call)
    # (the table offset is an argument baked into a capture: here every tick is captured anew, five captures of the same two calls)
    torch.cuda.synchronize()
    readings = []
    for tick in range(5):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            graphed.step_replay(table, ids, 20 * tick, 20)
            eyes[1].render_into(outs[1])
        g.replay()
        torch.cuda.synchronize()
        eager.step_replay(table, ids, 20 * tick, 20)
        eyes[0].render_into(outs[0])
        torch.cuda.synchronize()
        assert torch.equal(outs[0], outs[1]), tick
        for k in KEYS:
            assert torch.equal(eager.field(k), graphed.field(k)), (tick, k)
        readings.append(outs[0].clone())
    assert float((readings[-1] - readings[0]).abs().max()) > 0          # the flies moved: the views changed
    assert 0.05 < float(readings[-1].mean()) < 0.95
    # the handle-less entry: same readings; the plan is a copy of the buffers' contents at first use
    r = eyes[0]
    id_map, pale, inv_norm, plan = r.retina._device_constants(torch, eager.device)
    mine = [id_map.clone(), plan.clone(), pale.clone(), inv_norm.clone()]
    out2 = torch.zeros_like(outs[0])

    def legacy():
        _native.check(_native.lib().nmf_eye_render(
            eager._batch_h, ctypes.byref(r._params), r._spheres.data_ptr() if r._spheres is not None else None,
            r._cap_seg.data_ptr() if r._cap_seg is not None else None, r._cap_geom.data_ptr() if r._cap_geom is not None else None,
            mine[0].data_ptr(), mine[1].data_ptr(), mine[2].data_ptr(), mine[3].data_ptr(), r.retina.num_ommatidia, None, out2.data_ptr(), eager._stream()))
        torch.cuda.synchronize()

    legacy()
    assert torch.equal(out2, outs[0])
    mine[2].fill_(1)                      # every ommatidium "pale" — in the caller's buffer only
    out2.zero_(); legacy()
    assert torch.equal(out2, outs[0])     # same addresses: the cached plan (documented in include/nmf.h)
    mine[2] = mine[2].clone()             # another address: a new plan, now with the new contents
    out2.zero_(); legacy()
    assert not torch.equal(out2, outs[0])
    # a plan of another frame size is refused
    bad = ctypes.create_string_buffer(bytes(r._params), ctypes.sizeof(r._params))
    p2 = type(r._params).from_buffer(bad); p2.height = r._params.height // 2
    assert _native.lib().nmf_eye_render_planned(eager._batch_h, ctypes.byref(p2), r._plan_h, None, None, None, None, out2.data_ptr(), None) != 0
    assert b"plan" in _native.lib().nmf_last_error()
