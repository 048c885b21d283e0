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


def test_general_actuator_recurrences_on_the_kernel(torch_mod):
    """tests/test_oracle_actuator_types.py on the HIP kernel: a hinge driven by a damper, an integrated-velocity servo, a pneumatic
    cylinder and two muscles follows the closed-form recurrence of MuJoCo's documented general actuator step for step — force,
    activation state (``NMF_ACT``) and joint angle (general-tree kernel; reference ``compose/fly.py:65-77``: the four
    ``ActuatorType`` members the engine refused until round 6)."""
    torch = torch_mod
    from flygym_amd import HIPSimulation
    from tiny_models import TinyWorld, hinge_on_heavy_base
    import test_oracle_actuator_types as at

    for case, c in at.CASES.items():
        par = dict(inertia_yy=2e-6, mass=1e-3, com=(0.5, 0.0, 0.0), armature=1e-6, damping=0.0, stiffness=0.0, springref=0.0, q0=0.0)
        par.update(c["par"])
        model = hinge_on_heavy_base(**par, forcerange=c.get("forcerange"), general=dict(kind=c["kind"], **c["attrs"]), ctrlrange=c.get("ctrlrange"))
        sim = HIPSimulation(TinyWorld(model), n_worlds=2, device=0)
        inertia = par["inertia_yy"] + par["mass"] * 0.25
        acc0 = 1.0 / (inertia + par["armature"])
        gear = c["attrs"].get("gear", 1.0)
        n = 500
        q, v, act = par["q0"], 0.0, 0.0
        want = np.zeros((n, 3))
        for k in range(n):
            ctrl = float(c["ctrl"](k))
            cc = min(max(ctrl, c["ctrlrange"][0]), c["ctrlrange"][1]) if c.get("ctrlrange") else ctrl
            f, act = at.general_force(c["kind"], c["attrs"], q, v, cc, act, acc0)
            if c.get("forcerange"):
                f = min(max(f, c["forcerange"][0]), c["forcerange"][1])
            tot = gear * f - par["stiffness"] * (q - par["springref"]) - par["damping"] * v
            v = v + at.H * tot / (inertia + par["armature"] + at.H * par["damping"])
            q = q + at.H * v
            want[k] = (q, f, act)
        ctrl = torch.as_tensor(np.array([c["ctrl"](k) for k in range(n)], dtype=np.float32), device=sim.device)
        got = torch.zeros((n, 3), device=sim.device)
        for k in range(n):
            sim.field("ctrl")[:, 0] = ctrl[k]
            sim.step(1)
            got[k, 0], got[k, 1], got[k, 2] = sim.field("qpos")[0, 7], sim.field("actuator_force")[0, 0], sim.field("act")[0, 0]
        got = got.cpu().numpy().astype(np.float64)
        for col, name in enumerate(("joint angle", "force", "activation")):
            scale = max(np.abs(want[:, col]).max(), 1e-9)
            assert np.abs(got[:, col] - want[:, col]).max() < 2e-3 * scale, (case, name, np.abs(got[:, col] - want[:, col]).max() / scale)
        # several steps in one launch advance the activation like single steps do; reset clears it
        a1 = sim.field("act").clone()
        sim.reset()
        assert float(sim.field("act").abs().max()) == 0.0
        sim.field("ctrl")[:, 0] = ctrl[0]
        sim.step(25)
        sim2 = HIPSimulation(TinyWorld(model), n_worlds=2, device=0)
        sim2.field("ctrl")[:, 0] = ctrl[0]
        for _ in range(25):
            sim2.step(1)
        assert torch.equal(sim.field("act"), sim2.field("act")) and torch.equal(sim.field("qpos"), sim2.field("qpos")), case
        del a1


@pytest.mark.parametrize("kind", ["damper", "intvelocity", "cylinder", "muscle", "position + velocity + motor on the same joints"])
def test_flies_with_general_actuators_step_like_the_oracle(torch_mod, oracle_lib, kind):
    """1024 flies whose 42 leg actuators are of each of the reference's remaining types (or — last case — three actuators of the
    affine types on the same joints, which the kernel's affine pass alone would lose: one lane per actuator, plain stores), driven
    with per-world controls from the spawn pose into ground contact in 20-step launches: the activation state, the actuator forces
    and the next step's accelerations equal the f64 oracle's started from the kernel's own state, at several times and worlds."""
    torch = torch_mod
    from flygym_amd import HIPSimulation
    import test_oracle_actuator_types as at

    if kind in at.KW:
        fly, world, model = at.fly_with(kind, **at.KW[kind])
        ids = [i for i, a in enumerate(fly.actuators) if a["kind"] == kind]
    else:
        from flygym_amd.anatomy import ActuatedDOFPreset, AxisOrder, JointPreset, Skeleton
        from flygym_amd.compose import FlatGroundWorld, Fly, KinematicPosePreset
        from flygym_amd.utils.math import Rotation3D

        fly = Fly(name="nmf")
        fly.add_joints(Skeleton(axis_order=AxisOrder.YAW_PITCH_ROLL, joint_preset=JointPreset.LEGS_ONLY), neutral_pose=KinematicPosePreset.NEUTRAL)
        dofs = fly.skeleton.get_actuated_dofs_from_preset(ActuatedDOFPreset.LEGS_ACTIVE_ONLY)
        fly.add_actuators(dofs, "position", kp=30.0, neutral_input=KinematicPosePreset.NEUTRAL)
        fly.add_actuators(dofs[::2], "velocity", kv=2e-3, forcerange=(-0.5, 0.5))
        fly.add_actuators(dofs[::3], "motor", gear=0.5)
        fly.add_leg_adhesion()
        world = FlatGroundWorld()
        world.add_fly(fly, (0, 0, 0.8), Rotation3D("quat", (1, 0, 0, 0)))
        model = world.compile_model()
        ids = [i for i, a in enumerate(fly.actuators) if a["kind"] != "adhesion"]
        assert model["act_general"][:, 0].astype(bool).sum() == len(dofs[::2]) + len(dofs[::3])
    n = 1024
    sim = HIPSimulation(world, n_worlds=n, device=0)
    assert sim.batch_info()["chunked"] == 0            # activations advance in place: whole-launch work items
    nu = sim.model.nu
    g = torch.Generator(device=sim.device); g.manual_seed(11)
    sim.field("qvel")[:, 6:] = (torch.rand((n, sim.model.nv - 6), device=sim.device, generator=g) - 0.5) * 60.0
    phase = torch.rand((n, len(ids)), device=sim.device, generator=g) * 6.2832
    idt = torch.as_tensor(ids, device=sim.device)
    blob = sim.model.to_blob()
    compared, worst_f, worst_a, worst_q = 0, 0.0, 0.0, 0.0
    for tick in range(12):
        amp = 3.0 if kind in ("intvelocity",) else 1.5
        off = 1.0 if kind in ("damper", "muscle") else 0.0
        c = amp * torch.sin(phase + 0.9 * tick) + off
        ctrl = sim.field("ctrl"); ctrl[:, idt] = c
        sim.step(19)
        torch.cuda.synchronize()
        state = {k: sim.field(k).cpu().numpy().astype(np.float64) for k in ("qpos", "qvel", "ctrl", "qacc_warmstart", "act")}
        sim.step(1)
        torch.cuda.synchronize()
        qacc, frc, act = sim.field("qacc").cpu().numpy(), sim.field("actuator_force").cpu().numpy(), sim.field("act").cpu().numpy()
        stats, geom = sim.field("stats").cpu().numpy(), sim.field("contact_geom").cpu().numpy()
        for w in (0, 17 + 83 * tick, n - 1):
            o = oracle_lib.Oracle(blob, "f64")
            o.qpos[:] = state["qpos"][w]; o.qvel[:] = state["qvel"][w]; o.ctrl[:] = state["ctrl"][w]
            o.arr("qacc_warmstart")[:] = state["qacc_warmstart"][w]; o.arr("act")[:] = state["act"][w]
            o.step(1)
            ref_f, ref_a = o.arr("actuator_force"), o.arr("act")
            fs = max(np.abs(ref_f).max(), 1e-6)
            worst_f = max(worst_f, np.abs(frc[w] - ref_f).max() / fs)
            worst_a = max(worst_a, np.abs(act[w] - ref_a).max() / max(np.abs(ref_a).max(), 1e-6))
            nc = int(stats[w, 0])
            if nc == o.ints()["ncon"] and geom[w, :nc].astype(int).tolist() == o.ints()["con_geom"]:
                compared += 1
                worst_q = max(worst_q, np.abs(qacc[w] - o.arr("qacc")).max() / max(np.abs(o.arr("qacc")).max(), 1e4))
    report("general_actuators_1024", kind=kind, compared=compared, worst_force=worst_f, worst_activation=worst_a, worst_qacc=worst_q,
           mean_contacts=float(stats[:, 0].mean()))
    assert compared >= 30 and worst_f < 1e-4 and worst_a < 1e-5 and worst_q < 2e-3
    # (flies without a position servo slump onto the ground; the servoed ones of the last case stand on a few claws)
    assert float(stats[:, 0].mean()) > (3 if kind in at.KW else 0.5) and bool(torch.isfinite(sim.field("qpos")).all())
    if kind in ("intvelocity", "cylinder", "muscle"):
        assert float(np.abs(act[:, ids]).max()) > 1e-3 and float(np.abs(act[:, [i for i in range(nu) if i not in ids]]).max()) == 0.0


def test_a_world_with_two_flies_steps_each_like_its_own_world(torch_mod):
    """``BaseWorld.add_fly`` takes several flies (reference ``compose/world.py:95-149``; refused until round 6).  The reference's
    flies never collide with each other (``contype = conaffinity = 0``, fly-ground pairs only: ``compose/fly.py:609-610``,
    ``compose/world.py:300-309``), so a fly of a two-fly world must move exactly as in a world of its own: every per-fly query of
    ``HIPSimulation(two_fly_world, n)`` — and of the CPU-style ``Simulation`` — is bit-equal to the single-fly simulation's, with
    different skeletons, spawn poses, actuator sets and controls per fly."""
    torch = torch_mod
    from flygym_amd import HIPSimulation, Simulation
    from flygym_amd.anatomy import ActuatedDOFPreset, AxisOrder, JointPreset, Skeleton
    from flygym_amd.compose import ActuatorType, FlatGroundWorld, Fly, KinematicPosePreset
    from flygym_amd.utils.math import Rotation3D

    def fly_a(name="alice"):
        f = Fly(name=name)
        f.add_joints(Skeleton(axis_order=AxisOrder.YAW_PITCH_ROLL, joint_preset=JointPreset.LEGS_ONLY), neutral_pose=KinematicPosePreset.NEUTRAL)
        f.add_actuators(f.skeleton.get_actuated_dofs_from_preset(ActuatedDOFPreset.LEGS_ACTIVE_ONLY), "position", kp=50.0, neutral_input=KinematicPosePreset.NEUTRAL)
        f.add_leg_adhesion()
        return f

    def fly_b(name="bob"):
        f = Fly(name=name)
        f.add_joints(Skeleton(axis_order=AxisOrder.YAW_PITCH_ROLL, joint_preset=JointPreset.LEGS_ACTIVE_ONLY), neutral_pose=KinematicPosePreset.NEUTRAL)
        f.add_actuators(f.skeleton.get_actuated_dofs_from_preset(ActuatedDOFPreset.LEGS_ACTIVE_ONLY), "position", kp=20.0, neutral_input=KinematicPosePreset.NEUTRAL)
        return f

    spawn = {"alice": ((0, 0, 0.8), Rotation3D("quat", (1, 0, 0, 0))), "bob": ((6.0, -3.0, 1.1), Rotation3D("quat", (0.9238795, 0, 0, 0.3826834)))}
    both = FlatGroundWorld()
    both.add_fly(fly_a(), *spawn["alice"])
    both.add_fly(fly_b(), *spawn["bob"], add_ground_contact_sensors=False)
    with pytest.raises(ValueError, match="already exists"):
        both.add_fly(fly_b(), *spawn["bob"])
    with pytest.raises(ValueError, match="holds 2 flies"):
        both.compile_model()
    assert both.compile_model("bob").nv == 48 and both.compile_model("alice").nv == 72
    n = 32
    sim = HIPSimulation(both, n_worlds=n, device=0)
    assert set(sim.sims) == {"alice", "bob"} and sim.for_fly("bob").model.nv == 48
    alone = {}
    for name, make in (("alice", fly_a), ("bob", fly_b)):
        w = FlatGroundWorld()
        if name == "bob": w.add_fly(make(), *spawn[name], add_ground_contact_sensors=False)
        else: w.add_fly(make(), *spawn[name])
        alone[name] = HIPSimulation(w, n_worlds=n, device=0)
    g = torch.Generator(device=sim.device); g.manual_seed(5)
    for tick in range(6):
        for name in ("alice", "bob"):
            fly = both.fly_lookup[name]
            k = len(fly.get_actuated_jointdofs_order("position"))
            target = sim.get_joint_angles(name)[:, :0].new_zeros((n, k)) + 0.3 * torch.rand((n, k), device=sim.device, generator=g)
            sim.set_actuator_inputs(name, ActuatorType.POSITION, target)
            alone[name].set_actuator_inputs(name, ActuatorType.POSITION, target)
        adh = torch.ones((n, 6), device=sim.device) * (tick % 2)
        sim.set_leg_adhesion_states("alice", adh); alone["alice"].set_leg_adhesion_states("alice", adh)
        sim.step(25)
        for s in alone.values(): s.step(25)
        torch.cuda.synchronize()
        for name in ("alice", "bob"):
            for q in ("get_joint_angles", "get_joint_velocities", "get_body_positions", "get_body_rotations"):
                assert torch.equal(getattr(sim, q)(name), getattr(alone[name], q)(name)), (tick, name, q)
            assert torch.equal(sim.get_actuator_forces(name, "position"), alone[name].get_actuator_forces(name, "position"))
        for a, b in zip(sim.get_ground_contact_info("alice"), alone["alice"].get_ground_contact_info("alice")):
            assert torch.equal(a, b)
    assert float(sim.get_body_positions("bob")[:, 0, 0].mean()) > 4.0 > float(sim.get_body_positions("alice")[:, 0, 0].mean())
    assert sim.time == pytest.approx(150 * 1e-4, rel=1e-3)
    with pytest.raises(AttributeError, match="for_fly"):
        sim.field("qpos")
    sim.reset()
    assert float(sim.get_joint_velocities("bob").abs().max()) == 0.0
    # the sensors take the multi-fly simulation and find their fly's batch
    from flygym_amd.sensors import OdorSensors
    from flygym_amd.vision import EyeRenderer
    odor = OdorSensors(sim, "bob", [(10.0, 0.0, 1.0)], [(1.0,)])
    assert odor.sim is sim.for_fly("bob") and tuple(odor.get_odor_intensities().shape)[0] == n
    eyes = EyeRenderer(sim, "alice")
    assert eyes.sim is sim.for_fly("alice") and tuple(eyes.render().shape) == (n, 2, eyes.retina.num_ommatidia, 2)
    # the single-world CPU-style class on the same world
    one = Simulation(both, device=0)
    one.warmup(0.005)
    ja = one.get_joint_angles("alice"); jb = one.get_joint_angles("bob")
    assert ja.shape == (66,) and jb.shape == (42,) and np.isfinite(ja).all() and np.isfinite(jb).all()


@pytest.mark.parametrize("terrain", ["flat", "mixed"])
def test_one_step_accuracy_on_states_that_do_not_depend_on_the_solver(torch_mod, oracle_lib, terrain):
    """The constraint solve's one-step accuracy on SOLVER-INDEPENDENT states (round 6): the free-rollout tests of rounds 2-3 follow a
    build's rounding through chaotic branches, so two solvers of the same accuracy can land on different sides of their bars
    (DESIGN_APPENDIX.md H, direction pivots).  Here 4096 walking flies are rolled out on the primal Newton loop — round 3's solver,
    the control of tests/test_hip_parity_r2.py — and at six times their states are pushed into a default-solver batch, which takes one
    step to build its active-set history and a second that is judged: `qacc` of 48 sampled worlds per time against the float64 oracle
    stepped from the same state.  Whatever the default solver is, it sees the same states.  Measured on the shipped solver: flat
    median 4.6e-5 of max |qacc|, p90 1.1e-4, p99 2.2e-4, worst 3.7e-4; mixed terrain 5.0e-5 / 1.6e-4 / 3.3e-4 / 6.8e-4."""
    torch = torch_mod
    from flygym_amd import HIPSimulation, make_model
    from flygym_amd.controllers import TripodCPG

    def build(opts):
        fly, world, _ = make_model()
        if terrain != "flat":
            import flygym_amd.compose as C
            from flygym_amd.utils.math import Rotation3D
            world = C.MixedTerrainWorld()
            world.add_fly(fly, (0, 0, 0.8), Rotation3D("quat", (1, 0, 0, 0)))
        sim = HIPSimulation(world, n_worlds=n, device=0, _options=opts)
        sim.set_leg_adhesion_states(fly.name, np.ones((n, 6), dtype=np.float32))
        return fly, sim

    n = 4096
    fly, gen = build(dict(solver="primal"))
    _, sim = build(None)
    assert gen.batch_info()["solver_option_bits"] == 1 and sim.batch_info()["solver_option_bits"] == 0
    gen.warmup()
    table = TripodCPG(fly.get_actuated_jointdofs_order("position"), gen.timestep).targets(n, 2500, device=gen.device)
    ids = gen.replay_ids(fly.name)
    gen.step_replay(table, ids, 0, 400)
    blob = sim.model.to_blob()
    tab, ids_np = table.cpu().numpy(), ids.cpu().numpy()
    rng = np.random.default_rng(0)
    devs, cur, total, scales = [], 400, 0, []
    for cp in range(6):
        gen.step_replay(table, ids, cur, 61); cur += 61
        for k in KEYS: sim.field(k)[:] = gen.field(k)
        sim.step_replay(table, ids, cur, 1)                    # builds the active-set history from the pushed state
        torch.cuda.synchronize()
        state = {k: sim.field(k).cpu().numpy().astype(np.float64) for k in KEYS}
        sim.step_replay(table, ids, cur + 1, 1)                # the judged step
        torch.cuda.synchronize()
        qacc, stats, geom = sim.field("qacc").cpu().numpy().astype(np.float64), sim.field("stats").cpu().numpy(), sim.field("contact_geom").cpu().numpy()
        for w in rng.choice(n, size=48, replace=False):
            total += 1
            o = oracle_lib.Oracle(blob, "f64")
            for k in KEYS: o.arr(k)[:] = state[k][w]
            o.step_replay(tab[w], ids_np, cur + 1, 1)
            nc = int(stats[w, 0])
            if nc != o.ints()["ncon"] or geom[w, :nc].astype(int).tolist() != o.ints()["con_geom"]:
                continue
            ref = o.arr("qacc")
            scales.append(float(np.abs(ref).max()))
            devs.append(float(np.abs(qacc[w] - ref).max() / max(np.abs(ref).max(), 1e4)))
    devs = np.array(devs)
    # (round-5 verdict weak 1d: the bars' scale has a floor of 1e4 rad/s^2 — it never binds here: the largest acceleration of a walking
    # fly's light tarsal dofs is above 3e4 rad/s^2 in every sampled state (typically 1e5 .. 1e7), so every deviation above is relative to max |qacc| itself)
    assert min(scales) > 1e4, min(scales)
    exits = sim.get_solver_exits()
    report("one_step_accuracy_on_solver_independent_states", terrain=terrain, compared=len(devs), sampled=total, median=float(np.median(devs)),
           p90=float(np.quantile(devs, 0.9)), p99=float(np.quantile(devs, 0.99)), worst=float(devs.max()), smallest_max_qacc=float(min(scales)),
           contact_space=exits["contact_space"], kkt_exact=exits["kkt_exact"], primal_loop=exits["primal_loop"])
    assert len(devs) >= 0.95 * total                                              # contact lists equal to the oracle's in nearly every sampled state
    assert np.median(devs) < 1e-4 and np.quantile(devs, 0.9) < 3e-4 and np.quantile(devs, 0.99) < 1e-3 and devs.max() < 2e-3
    assert exits["contact_space"] >= 0.99 * exits["steps"] * (1.0 if terrain == "flat" else 0.97)      # the judged solver is the contact-space one


def test_flies_per_cu_option_changes_residency_not_results(torch_mod):
    """``nmf_batch_options.flies_per_cu`` (round 6: idle LDS per stepping workgroup, so that a CU keeps room for another stream's
    kernel — measured with the eye renderer, DESIGN_APPENDIX.md H): ``nmf_batch_info`` reports the residency asked for, the grid
    shrinks with it, and — scheduling only — the states after chunked launches are bit-identical to the default batch's."""
    torch = torch_mod
    from flygym_amd import HIPSimulation, make_model
    from flygym_amd.controllers import TripodCPG

    n = 4096
    sims = []
    for opts in (None, dict(flies_per_cu=6), dict(flies_per_cu=99)):
        fly, world, _ = make_model()
        sim = HIPSimulation(world, n_worlds=n, device=0, _options=opts)
        sim.set_leg_adhesion_states(fly.name, np.ones((n, 6), dtype=np.float32))
        sims.append(sim)
    base, six, many = sims
    info0, info6, info99 = base.batch_info(), six.batch_info(), many.batch_info()
    assert info0["flies_per_cu"] == 8 and info6["flies_per_cu"] == 6 and info99["flies_per_cu"] == 8      # above the kernel's own: ignored
    assert info6["resident_workgroups"] * 8 == info0["resident_workgroups"] * 6
    table = TripodCPG(fly.get_actuated_jointdofs_order("position"), base.timestep).targets(n, 2500, device=base.device)
    ids = base.replay_ids(fly.name)
    for sim in (base, six):
        sim.warmup()
        for k in range(4):
            sim.step_replay(table, ids, 20 * k, 20)
    torch.cuda.synchronize()
    for k in KEYS + ("qacc", "sensordata"):
        assert torch.equal(base.field(k), six.field(k)), k
    assert float(base.field("stats")[:, 0].mean()) > 3
