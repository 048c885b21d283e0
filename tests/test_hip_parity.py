"""Parity of the HIP engine (through the C ABI / HIPSimulation) with the CPU oracle, on MI355X.

Tolerances: the engine computes in float32; the oracle in float64 (and float32 for the
integer-exact comparisons).  One physics step from identical states must agree to float32
rounding of a stiff system: accelerations to 2e-3 relative (of the largest |qacc|), next-step
positions to 1e-5 mm/rad, velocities to 5e-3 relative.  Contact counts and contact geom ids must be
bit-exact against the float32 oracle.  Rollouts (chaotic, contact-rich) are compared over a few
hundred steps with the looser bounds written in each test.
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


@pytest.fixture(scope="module")
def sim8(torch_mod):
    from flygym_amd import HIPSimulation, make_model

    fly, world, _ = make_model()          # its own world: HIPSimulation strips the noslip option in place
    sim = HIPSimulation(world, n_worlds=8, device=0)
    return fly, sim


def _oracles(oracle_lib, sim):
    blob = sim.model.to_blob()
    return oracle_lib.Oracle(blob, "f64"), oracle_lib.Oracle(blob, "f32")


def _push_state(sim, torch, qpos, qvel, ctrl, ws):
    sim.field("qpos")[:] = torch.as_tensor(qpos, dtype=torch.float32, device=sim.device)
    sim.field("qvel")[:] = torch.as_tensor(qvel, dtype=torch.float32, device=sim.device)
    sim.field("ctrl")[:] = torch.as_tensor(ctrl, dtype=torch.float32, device=sim.device)
    sim.field("qacc_warmstart")[:] = torch.as_tensor(ws, dtype=torch.float32, device=sim.device)


def _sample_states(oracle_lib, sim, n, seed):
    """States along a walking trajectory (contact-rich) with random velocity perturbations."""
    rng = np.random.default_rng(seed)
    o, _ = _oracles(oracle_lib, sim)
    o.ctrl[42:] = 1.0
    o.step(400)
    states = []
    for k in range(n):
        o.ctrl[:42] = sim.model["key_ctrl"][:42] + rng.normal(0, 0.25, 42)
        o.step(60)
        qvel = o.qvel.copy() + rng.normal(0, 0.5, o.nv) * (k % 2)
        states.append((o.qpos.copy(), qvel, o.ctrl.copy(), o.arr("qacc_warmstart").copy()))
    return states


def test_single_step_parity(torch_mod, sim8, oracle_lib):
    torch = torch_mod
    fly, sim = sim8
    states = _sample_states(oracle_lib, sim, 8, seed=11)
    _push_state(sim, torch, *[np.stack([s[i] for s in states]) for i in range(4)])
    sim.step(1)
    torch.cuda.synchronize()
    qacc = sim.field("qacc").cpu().numpy()
    qpos = sim.field("qpos").cpu().numpy()
    qvel = sim.field("qvel").cpu().numpy()
    stats = sim.field("stats").cpu().numpy()
    frc = sim.field("actuator_force").cpu().numpy()
    sens = sim.field("sensordata").cpu().numpy()
    ncon_seen = 0
    for w, (q0, v0, c0, ws0) in enumerate(states):
        o64, o32 = _oracles(oracle_lib, sim)
        for o in (o64, o32):
            o.qpos[:] = q0; o.qvel[:] = v0; o.ctrl[:] = c0; o.arr("qacc_warmstart")[:] = ws0
            o.step(1)
        assert int(stats[w, 0]) == o32.ints()["ncon"] == o64.ints()["ncon"]     # integer parity
        ncon_seen += int(stats[w, 0])
        scale = np.abs(o64.arr("qacc")).max()
        assert np.abs(qacc[w] - o64.arr("qacc")).max() < 2e-3 * scale
        assert np.abs(qpos[w] - o64.qpos).max() < 1e-5
        assert np.abs(qvel[w] - o64.qvel).max() < 5e-3 * max(1.0, np.abs(o64.qvel).max())
        np.testing.assert_allclose(frc[w], o64.arr("actuator_force"), rtol=1e-4, atol=1e-4)
        so = o64.arr("sensordata").reshape(6, 16)
        sh = sens[w].reshape(6, 16)
        np.testing.assert_array_equal(sh[:, 0], so[:, 0])
        np.testing.assert_allclose(sh[:, 1:4], so[:, 1:4], rtol=5e-3, atol=5e-3 * np.abs(so[:, 1:4]).max())
        np.testing.assert_allclose(sh[:, 7:10], so[:, 7:10], atol=1e-4)
    assert ncon_seen >= 24            # the sampled states really are in contact


def test_rollout_parity_and_reference_invariants(torch_mod, sim8, oracle_lib):
    torch = torch_mod
    fly, sim = sim8
    sim.reset()
    assert sim.time == pytest.approx(0.0)
    neutral = np.array([fly.jointdof_to_neutralangle[d] for d in fly.get_jointdofs_order()], dtype=np.float32)
    np.testing.assert_allclose(sim.get_joint_angles(fly.name).cpu().numpy(), np.tile(neutral, (8, 1)), atol=1e-6)
    assert float(sim.get_joint_velocities(fly.name).abs().max()) == 0.0
    o64, _ = _oracles(oracle_lib, sim)
    np.testing.assert_allclose(sim.get_body_positions(fly.name).cpu().numpy()[0],
                               o64.arr("seg_xpos").reshape(69, 3), atol=2e-6)
    sim.set_leg_adhesion_states(fly.name, np.ones((8, 6), dtype=np.float32))
    o64.ctrl[42:] = 1.0
    for k in range(6):
        sim.step(50)
        o64.step(50)
        q = sim.field("qpos").cpu().numpy()
        assert np.abs(q - o64.qpos[None]).max() < 5e-5, f"after {50 * (k + 1)} steps"
        assert np.abs(q - q[0:1]).max() == 0.0                      # identical worlds stay identical
    assert sim.time == pytest.approx(300 * 1e-4, rel=1e-3)
    quats = sim.get_body_rotations(fly.name).cpu().numpy()
    assert quats.shape == (8, 69, 4)
    np.testing.assert_allclose(np.linalg.norm(quats, axis=2), 1.0, atol=1e-5)
    np.testing.assert_allclose(sim.get_body_positions(fly.name).cpu().numpy()[3],
                               o64.arr("seg_xpos").reshape(69, 3), atol=1e-4)
    active, force, torque, pos, normal, tangent = sim.get_ground_contact_info(fly.name)
    assert active.shape == (8, 6) and force.shape == (8, 6, 3)
    so = o64.arr("sensordata").reshape(6, 16)
    np.testing.assert_array_equal(active.cpu().numpy()[0], so[:, 0])
    np.testing.assert_allclose(force.cpu().numpy()[0], so[:, 1:4], rtol=2e-2, atol=2e-2)
    weight = sim.model["body_mass"].sum() * 9810.0
    assert float(force[0, :, 2].sum()) == pytest.approx(weight + 6.0, rel=0.1)


def test_control_inputs_and_getters(torch_mod, sim8):
    torch = torch_mod
    from flygym_amd.compose import ActuatorType

    fly, sim = sim8
    sim.reset()
    n_dofs = sim.get_joint_angles(fly.name).shape[1]
    assert n_dofs == 66
    before = sim.get_joint_angles(fly.name).cpu().numpy().copy()
    rng = np.random.default_rng(5)
    inputs = rng.normal(0, 0.5, (8, 42)).astype(np.float32)
    sim.set_actuator_inputs(fly.name, ActuatorType.POSITION, inputs)
    ctrl = sim.field("ctrl").cpu().numpy()
    np.testing.assert_array_equal(ctrl[:, :42], inputs)                 # scatter is exact
    sim.set_leg_adhesion_states(fly.name, torch.full((8, 6), 3.0, device=sim.device))
    np.testing.assert_array_equal(sim.field("ctrl").cpu().numpy()[:, 42:], 3.0)
    with pytest.raises(ValueError):
        sim.set_actuator_inputs(fly.name, ActuatorType.POSITION, np.zeros((8, 41), dtype=np.float32))
    with pytest.raises(ValueError):
        sim.set_actuator_inputs(fly.name, ActuatorType.POSITION, np.zeros((7, 42), dtype=np.float32))
    with pytest.raises(ValueError):
        sim.set_leg_adhesion_states(fly.name, np.ones((8, 5), dtype=np.float32))
    # a wider array: its leading columns are used, as the reference's GPU class does (warp/simulation.py:236-258)
    wide = np.concatenate([inputs, np.full((8, 24), 9.0, dtype=np.float32)], axis=1)
    sim.set_actuator_inputs(fly.name, ActuatorType.POSITION, 2 * wide)
    np.testing.assert_array_equal(sim.field("ctrl").cpu().numpy()[:, :42], 2 * inputs)
    np.testing.assert_array_equal(sim.field("ctrl").cpu().numpy()[:, 42:], 3.0)
    sim.set_actuator_inputs(fly.name, ActuatorType.POSITION, inputs)
    for _ in range(50):
        sim.step()
    after = sim.get_joint_angles(fly.name).cpu().numpy()
    assert not np.allclose(before, after, atol=1e-4)                      # reference test_control_inputs_affect...
    assert np.abs(after - after[0]).max() > 1e-3                         # different inputs -> different worlds
    forces = sim.get_actuator_forces(fly.name, ActuatorType.POSITION)
    assert forces.shape == (8, 42) and float(forces.abs().max()) <= 30.0 + 1e-4    # forcerange clamp
    # gathers equal plain indexing of the raw arrays
    np.testing.assert_array_equal(sim.get_joint_velocities(fly.name).cpu().numpy(), sim.field("qvel").cpu().numpy()[:, 6:])
    np.testing.assert_array_equal(sim.get_body_rotations(fly.name).cpu().numpy().reshape(8, -1),
                                  sim.field("seg_xquat").cpu().numpy())


def test_in_kernel_replay_equals_per_step_scatter(torch_mod, bench_model):
    torch = torch_mod
    from flygym_amd import HIPSimulation
    from flygym_amd.compose import ActuatorType
    from flygym_amd.replay import ReplayTargetData

    fly, world, _ = bench_model
    order = fly.get_actuated_jointdofs_order(ActuatorType.POSITION)
    table = ReplayTargetData(1e-4, order).make_target_angles_all_worlds(6, 1000)
    a = HIPSimulation(world, n_worlds=6, device=0)
    b = HIPSimulation(world, n_worlds=6, device=0)
    tdev = torch.as_tensor(table, device=a.device)
    ids = a._ids_by_fly[fly.name]["actuators"][ActuatorType.POSITION]
    a.step(100); b.step(100)
    a.step_replay(tdev, ids, 990, 40)                                   # wraps around the table end
    for s in range(40):
        b.set_actuator_inputs(fly.name, ActuatorType.POSITION, tdev[:, (990 + s) % 1000, :])
        b.step()
    torch.cuda.synchronize()
    assert torch.equal(a.field("qpos"), b.field("qpos")) and torch.equal(a.field("qvel"), b.field("qvel"))


def test_full_size_batch_properties(torch_mod, bench_model):
    """BASELINE size (4096 worlds): finite, deterministic across launches, worlds independent."""
    torch = torch_mod
    from flygym_amd import HIPSimulation
    from flygym_amd.compose import ActuatorType
    from flygym_amd.replay import ReplayTargetData

    fly, world, _ = bench_model
    n = 4096
    order = fly.get_actuated_jointdofs_order(ActuatorType.POSITION)
    table = torch.as_tensor(ReplayTargetData(1e-4, order).make_target_angles_all_worlds(n, 1000), device="cuda:0")
    runs = []
    for rep in range(2):
        sim = HIPSimulation(world, n_worlds=n, device=0)
        ids = sim._ids_by_fly[fly.name]["actuators"][ActuatorType.POSITION]
        sim.set_leg_adhesion_states(fly.name, np.ones((n, 6), dtype=np.float32))
        sim.step(300)
        if rep == 1:                          # perturb one world only
            sim.field("qvel")[1234, 6:] += 5.0
        sim.step_replay(table, ids, 0, 150)
        torch.cuda.synchronize()
        runs.append((sim.field("qpos").clone(), sim.field("stats").clone()))
        del sim
    q0, q1 = runs[0][0], runs[1][0]
    assert bool(torch.isfinite(q0).all())
    same = (q0 == q1).all(dim=1)
    assert int((~same).sum()) == 1 and not bool(same[1234])             # bitwise determinism + independence
    # worlds w and w+20 replay the same partition from the same state -> identical
    assert torch.equal(q0[5], q0[25]) and not torch.equal(q0[5], q0[6])
    assert float(runs[0][1][:, 2].sum()) == 0.0                         # no contact overflow
    assert float(runs[0][1][:, 0].mean()) > 2.0                         # walking: legs on the ground


def test_legs_active_only_skeleton_parity(torch_mod, oracle_lib):
    """The second compiled topology (LEGS_ACTIVE_ONLY: 6 x (3,2,1,1) hinges, nv 48) with all-capsule
    geometry and non-default contact parameters: 200-step rollout against the float64 oracle."""
    torch = torch_mod
    from flygym_amd import HIPSimulation, anatomy as A
    from flygym_amd.compose import (ActuatorType, ContactParams, FlatGroundWorld, Fly, GeomFittingOption,
                                    KinematicPosePreset)
    from flygym_amd.utils.math import Rotation3D

    fly = Fly(name="active", geom_fitting_option=GeomFittingOption.ALL_TO_CAPSULES)
    sk = A.Skeleton(axis_order=A.AxisOrder.YAW_PITCH_ROLL, joint_preset=A.JointPreset.LEGS_ACTIVE_ONLY)
    fly.add_joints(sk, neutral_pose=KinematicPosePreset.NEUTRAL)
    fly.add_actuators(sk.get_actuated_dofs_from_preset("all"), ActuatorType.POSITION, kp=40.0,
                      neutral_input=KinematicPosePreset.NEUTRAL)
    fly.add_leg_adhesion(gain=2.0)
    world = FlatGroundWorld()
    world.add_fly(fly, (0.0, 0.0, 0.9), Rotation3D("quat", (0.9987503, 0.0, 0.0499792, 0.0)),
                  ground_contact_params=ContactParams(sliding_friction=1.5))
    sim = HIPSimulation(world, n_worlds=3, device=0)
    assert sim.model.nv == 48 and sim.get_joint_angles(fly.name).shape == (3, 42)
    o = oracle_lib.Oracle(sim.model.to_blob(), "f64")
    rng = np.random.default_rng(3)
    targets = sim.model["key_ctrl"][:42] + rng.normal(0, 0.2, 42)
    sim.set_actuator_inputs(fly.name, ActuatorType.POSITION, np.tile(targets.astype(np.float32), (3, 1)))
    o.ctrl[:42] = targets.astype(np.float32)
    for k in range(4):
        sim.step(50); o.step(50)
        q = sim.field("qpos").cpu().numpy()
        assert np.abs(q - o.qpos[None]).max() < 2e-4, f"after {50 * (k + 1)} steps"
    assert int(sim.field("stats")[0, 0].item()) == o.ints()["ncon"] > 0


@pytest.mark.parametrize("preset", ["LEGS_ONLY", "ALL_BIOLOGICAL"])
def test_tethered_world_parity(torch_mod, oracle_lib, preset):
    """TetheredWorld (reference compose/world.py:334-366): the root is held by a soft 6-row weld.  The fixture
    mirrors the reference's tests/conftest.py (LEGS_ONLY, YPR, 42 position actuators kp 50, adhesion, spawn z 1.5);
    ALL_BIOLOGICAL takes the same weld through the general-tree kernel."""
    torch = torch_mod
    from flygym_amd import HIPSimulation, anatomy as A
    from flygym_amd.compose import ActuatorType, Fly, KinematicPosePreset, TetheredWorld
    from flygym_amd.utils.math import Rotation3D

    fly = Fly(name="t")
    fly.add_joints(A.Skeleton(axis_order=A.AxisOrder.YAW_PITCH_ROLL, joint_preset=getattr(A.JointPreset, preset)),
                   neutral_pose=KinematicPosePreset.NEUTRAL)
    sk = A.Skeleton(axis_order=A.AxisOrder.YAW_PITCH_ROLL, joint_preset=A.JointPreset.LEGS_ONLY)
    fly.add_actuators(sk.get_actuated_dofs_from_preset("legs_active_only"), ActuatorType.POSITION, kp=50.0,
                      neutral_input=KinematicPosePreset.NEUTRAL)
    fly.add_leg_adhesion()
    world = TetheredWorld()
    world.add_fly(fly, (0, 0, 1.5), Rotation3D("quat", (1, 0, 0, 0)))
    sim = HIPSimulation(world, n_worlds=3, device=0)
    o = oracle_lib.Oracle(sim.model.to_blob(), "f64")
    rng = np.random.default_rng(4)
    targets = (sim.model["key_ctrl"][:42] + rng.normal(0, 0.4, 42)).astype(np.float32)
    sim.set_actuator_inputs(fly.name, ActuatorType.POSITION, np.tile(targets, (3, 1)))
    o.ctrl[:42] = targets
    for k in range(4):
        sim.step(75); o.step(75)
        q = sim.field("qpos").cpu().numpy()
        assert np.abs(q - o.qpos[None]).max() < 5e-5, f"after {75 * (k + 1)} steps"
    # the tether holds: root within a few 1e-5 mm / rad of its spawn pose, legs have moved
    np.testing.assert_allclose(q[0, :7], [0, 0, 1.5, 1, 0, 0, 0], atol=2e-4)
    assert np.abs(q[0, 7:] - sim.model["key_qpos"][7:]).max() > 0.1
    assert sim.time == pytest.approx(300e-4, rel=1e-3)
    assert int(sim.field("stats")[0, 0].item()) == 0                      # no ground, no contacts
    with pytest.raises(ValueError):
        sim.get_ground_contact_info(fly.name)                              # no contact sensors in this world


@pytest.mark.parametrize("world_cls", ["GappedTerrainWorld", "BlocksTerrainWorld", "MixedTerrainWorld"])
def test_terrain_worlds_parity(torch_mod, oracle_lib, world_cls):
    """BASELINE config 4/5 terrains (build-defined height maps): dropping onto them and CPG walking across them, HIP vs
    the float64 oracle in re-synchronised 20-step segments."""
    torch = torch_mod
    import flygym_amd.compose as C
    from flygym_amd import HIPSimulation, anatomy as A
    from flygym_amd.controllers import TripodCPG
    from flygym_amd.utils.math import Rotation3D

    fly = C.Fly(name="t")
    sk = A.Skeleton(axis_order=A.AxisOrder.YAW_PITCH_ROLL, joint_preset=A.JointPreset.LEGS_ONLY)
    fly.add_joints(sk, neutral_pose=C.KinematicPosePreset.NEUTRAL)
    fly.add_actuators(sk.get_actuated_dofs_from_preset("legs_active_only"), C.ActuatorType.POSITION, kp=50.0,
                      neutral_input=C.KinematicPosePreset.NEUTRAL)
    fly.add_leg_adhesion()
    world = getattr(C, world_cls)()
    world.add_fly(fly, (0.3, 0.2, 0.8), Rotation3D("quat", (1, 0, 0, 0)))
    sim = HIPSimulation(world, n_worlds=4, device=0)
    o = oracle_lib.Oracle(sim.model.to_blob(), "f64")
    order = fly.get_actuated_jointdofs_order(C.ActuatorType.POSITION)
    table = TripodCPG(order, 1e-4).targets(1, 2500)
    tdev = torch.as_tensor(np.repeat(table, 4, axis=0), device=sim.device)
    ids = sim._ids_by_fly[fly.name]["actuators"][C.ActuatorType.POSITION]
    sim.set_leg_adhesion_states(fly.name, np.ones((4, 6), dtype=np.float32))
    o.ctrl[42:] = 1.0
    # A height map is discontinuous: when a hull vertex sits on a block edge, float32 and float64 (or two float32
    # evaluation orders) may put it on different sides, and a contact-rich trajectory separates from there.  So the
    # comparison is re-synchronised: every 20 steps the float64 oracle's state is pushed into the engine, both advance
    # 20 steps under the same controls, and the positions are compared.  Round 3: every segment runs step by step with
    # the contact lists of the engine and of both oracles recorded, and a segment that ends further than rounding from
    # both oracles must be EXPLAINED (tests/resync.py): either the float32 and float64 oracles themselves pick different
    # contact sets inside it, or the float64 oracle reproduces the engine's contact set from the engine's own state
    # under a rounding-sized perturbation (a genuine near-tie).  No percentage of unexplained segments is tolerated.
    from resync import Resync

    o32 = oracle_lib.Oracle(sim.model.to_blob(), "f32")
    o32.ctrl[42:] = 1.0
    rs = Resync(sim, torch, oracle_lib, o, o32, tol=2e-5)
    ids_np = np.arange(42, dtype=np.int32)
    for k in range(20):                                    # the drop onto the terrain and settling (400 steps)
        rs.segment(20, lambda i: sim.step(1), lambda orc, i: orc.step(1), label=f"drop {k}")
    for k in range(15):                                    # CPG walking across it (300 steps)
        rs.segment(20, lambda i, k=k: sim.step_replay(tdev, ids, 20 * k + i, 1),
                   lambda orc, i, k=k: orc.step_replay(table[0], ids_np, 20 * k + i, 1), label=f"walk {k}")
    kinds = rs.summary()
    errs64 = np.array([r["err_base"] for r in rs.records])
    assert not rs.violations, f"{world_cls}: {kinds}\n" + "\n".join(rs.violations)
    assert kinds["rounding"] >= 0.8 * len(rs.records), f"{world_cls}: {kinds}"     # events stay the exception
    assert errs64.max() < 5e-3, f"{world_cls}: worst segment {errs64.max():.2e}"
    assert max(r["ncon_end"] for r in rs.records) >= 3
    assert np.isfinite(sim.field("qpos").cpu().numpy()).all()
    assert o.qpos[0] > 0.3 + 0.1                           # the fly actually walked forward over the terrain


def test_single_world_and_launch_splitting(torch_mod, bench_model, oracle_lib):
    """Edge cases: n_worlds = 1; a fly dropped low with all leg targets at zero (body hulls touch the ground: many
    hull contacts); the same steps split into different launch sizes are bitwise identical."""
    torch = torch_mod
    from flygym_amd import HIPSimulation
    from flygym_amd.compose import ActuatorType

    fly, world, _ = bench_model
    a = HIPSimulation(world, n_worlds=1, device=0)
    b = HIPSimulation(world, n_worlds=1, device=0)
    zeros = np.zeros((1, 42), dtype=np.float32)
    for sim in (a, b):
        sim.set_actuator_inputs(fly.name, ActuatorType.POSITION, zeros)      # fold the legs
        sim.field("qpos")[:, 2] = 0.3
    a.step(600)
    for _ in range(10):
        b.step(60)
    torch.cuda.synchronize()
    a.step(60)
    b.step(0 + 60)
    # launch splitting: 600 + 60 in (1 + 1) launches vs (10 + 1) launches of 60
    assert torch.equal(a.field("qpos"), b.field("qpos")) and torch.equal(a.field("qvel"), b.field("qvel"))
    stats = a.field("stats").cpu().numpy()[0]
    assert int(stats[0]) >= 1 and stats[2] == 0
    assert np.isfinite(a.field("qpos").cpu().numpy()).all()
    # parity in this regime, re-synchronised to the float32 oracle every 20 steps: a contact that crosses the margin
    # inside a segment may switch on one step apart (float32 rounding of a distance against the margin) and kick the
    # stiff contact differently; everything else must agree to rounding.  Round 3: each segment that does not is
    # classified step by step (tests/resync.py) — the two oracles disagree on the contact set, or the float64 oracle
    # reproduces the engine's set from the engine's own state — and none may stay unexplained.
    from resync import Resync

    o = oracle_lib.Oracle(a.model.to_blob(), "f32")
    o64 = oracle_lib.Oracle(a.model.to_blob(), "f64")
    o.ctrl[:42] = 0.0
    o.qpos[2] = 0.3
    rs = Resync(a, torch, oracle_lib, o, o64, tol=5e-6)
    for k in range(33):
        rs.segment(20, lambda i: a.step(1), lambda orc, i: orc.step(1), label=f"belly {k}")
    kinds = rs.summary()
    assert not rs.violations, f"{kinds}\n" + "\n".join(rs.violations)
    assert kinds["rounding"] >= 0.8 * len(rs.records), kinds
    assert max(r["err_base"] for r in rs.records) < 5e-3
    assert max(r["ncon_end"] for r in rs.records) >= 2


def test_reset_worlds_mask(torch_mod, bench_model):
    """Per-world episode reset (nmf_reset_worlds): masked worlds go back to the keyframe with their clock at zero and
    then follow the same trajectory as a freshly reset batch, bit for bit; unmasked worlds are not touched."""
    torch = torch_mod
    from flygym_amd import HIPSimulation

    fly, world, _ = bench_model
    sim = HIPSimulation(world, n_worlds=6, device=0)
    ref = HIPSimulation(world, n_worlds=6, device=0)
    for s in (sim, ref):
        s.set_leg_adhesion_states(fly.name, np.ones((6, 6), dtype=np.float32))
    sim.step(300)
    before = {k: sim.field(k).clone() for k in ("qpos", "qvel", "qacc_warmstart", "time", "seg_xpos")}
    mask = np.array([1, 0, 0, 1, 0, 1], dtype=bool)
    sim.reset_worlds(mask)
    on, off = torch.as_tensor(mask, device=sim.device), torch.as_tensor(~mask, device=sim.device)
    for k in before:
        assert torch.equal(sim.field(k)[off], before[k][off]), k
        assert torch.equal(sim.field(k)[on], ref.field(k)[on]), k
    assert float(sim.field("time")[0]) == 0.0 and abs(float(sim.field("time")[1]) - 0.03) < 1e-6
    # adhesion controls are part of the keyframe reset; put them back on, then step both batches
    sim.set_leg_adhesion_states(fly.name, np.ones((6, 6), dtype=np.float32))
    sim.step(200); ref.step(200)
    for k in ("qpos", "qvel", "time"):
        assert torch.equal(sim.field(k)[on], ref.field(k)[on]), k
    with pytest.raises(ValueError):
        sim.reset_worlds(np.ones(5, dtype=bool))


def test_cpg_adhesion_replay_parity(torch_mod, bench_model, oracle_lib):
    """BASELINE config 5 control: a 48-column table (42 joint targets + 6 gait-driven adhesion controls) replayed inside
    the kernel follows the oracle."""
    torch = torch_mod
    from flygym_amd import HIPSimulation
    from flygym_amd.controllers import TripodCPG

    fly, world, _ = bench_model
    sim = HIPSimulation(world, n_worlds=4, device=0)
    o, o32 = _oracles(oracle_lib, sim)
    cpg = TripodCPG(fly.get_actuated_jointdofs_order("position"), 1e-4)
    adhesion = (cpg.stance_bins(sim.model, fly), 20.0, 1.0)
    table = cpg.targets(4, 2500, adhesion=adhesion)
    tdev = cpg.targets(4, 2500, device=sim.device, adhesion=adhesion)
    np.testing.assert_allclose(tdev.cpu().numpy()[..., :42], table[..., :42], atol=2e-6)
    tdev = torch.as_tensor(table, device=sim.device)           # identical inputs for the comparison
    ids = sim.replay_ids(fly.name, with_adhesion=True)
    assert ids.numel() == 48
    sim.set_leg_adhesion_states(fly.name, np.ones((4, 6), dtype=np.float32))
    for orc in (o, o32):
        orc.ctrl[42:] = 1.0
        orc.step(300)
    sim.step(300)
    ids_np = ids.cpu().numpy()
    for k in range(3):
        sim.step_replay(tdev, ids, 100 * k, 100)
        for orc in (o, o32):
            orc.step_replay(table[2], ids_np, 100 * k, 100)
        q = sim.field("qpos").cpu().numpy()[2]
        assert np.abs(q - o.qpos).max() < 5e-5, f"after {300 + 100 * (k + 1)} steps"
    c = sim.field("ctrl").cpu().numpy()
    np.testing.assert_array_equal(c[2][ids_np], table[2, 299])
    f = sim.get_actuator_forces(fly.name, "position").cpu().numpy()
    assert np.isfinite(f).all()


def test_step_and_setters_are_graph_capturable(torch_mod, bench_model):
    """SURVEY §8(b) threading: setters and step are pure stream-ordered device work, so the reference's captured loop
    {set_actuator_inputs, step} (time_gpu_simulation.py:137-146) can be a hipGraph.  Replaying the graph must give the
    same states as the eager calls."""
    torch = torch_mod
    from flygym_amd import HIPSimulation

    fly, world, _ = bench_model
    n = 16
    eager = HIPSimulation(world, n_worlds=n, device=0)
    graphed = HIPSimulation(world, n_worlds=n, device=0)
    g = torch.Generator(device="cuda").manual_seed(3)
    base = eager.field("ctrl")[:, :42].clone()
    targets = [base + 0.05 * torch.randn((n, 42), device="cuda", generator=g) for _ in range(6)]
    adh = torch.ones((n, 6), device="cuda")
    for sim in (eager, graphed):
        sim.set_leg_adhesion_states(fly.name, adh)
        sim.step(50)
    for tg in targets:
        eager.set_actuator_inputs(fly.name, "position", tg)
        eager.step(5)
    static_in = targets[0].clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):                      # warm the capture stream once, then rewind the state
        graphed.set_actuator_inputs(fly.name, "position", static_in)
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        graphed.set_actuator_inputs(fly.name, "position", static_in)
        graphed.step(5)
    # the capture itself does not execute; state is still the post-warm-up one
    for tg in targets:
        static_in.copy_(tg)
        graph.replay()
    torch.cuda.synchronize()
    for k in ("qpos", "qvel", "ctrl"):
        assert torch.equal(eager.field(k), graphed.field(k)), k


def test_single_world_simulation_mirrors_the_cpu_class(torch_mod, bench_model, oracle_lib):
    """flygym_amd.Simulation: the reference's CPU ``Simulation`` surface (unbatched numpy) over the HIP engine —
    the reference's own invariants (tests/core/test_simulation.py: time advance, unit quaternions, zero velocity at
    reset, wrong-length errors) plus parity of a short driven rollout with the oracle (BASELINE config 1) — on the CPU
    class's engine: Newton + 5 noslip sweeps in kernel and oracle (round-5 verdict 1a: a world shared with a HIPSimulation
    had lost the option, and the test compared the batched flavours)."""
    from flygym_amd import Simulation
    from flygym_amd.replay import ReplayTargetData

    fly, world, _ = bench_model                  # function-scoped: a world no HIPSimulation has stripped
    assert world.noslip_iterations == 5          # mujoco_globals.yaml:15 — the CPU class's engine keeps the pass
    sim = Simulation(world, device=0)
    assert world.noslip_iterations == 5 and sim.batch.batch_info()["noslip_iterations"] == 5
    assert sim.time == 0.0 and abs(sim.timestep - 1e-4) < 1e-12
    q0 = sim.get_joint_angles(fly.name)
    assert q0.shape == (66,) and q0.dtype == np.float64
    assert sim.get_joint_velocities(fly.name).shape == (66,) and not sim.get_joint_velocities(fly.name).any()
    assert sim.get_body_positions(fly.name).shape == (69, 3)
    quat = sim.get_body_rotations(fly.name)
    assert quat.shape == (69, 4) and np.allclose(np.linalg.norm(quat, axis=1), 1.0, atol=1e-5)
    assert sim.get_actuator_forces(fly.name, "position").shape == (42,)
    order = fly.get_actuated_jointdofs_order("position")
    targets = ReplayTargetData(sim.timestep, order).make_target_angles_all_worlds(1, 200)[0]
    o = oracle_lib.Oracle(sim.batch.model.to_blob(), "f64", cpu_flavour=True)      # the oracle runs the pass too
    sim.set_leg_adhesion_states(fly.name, np.ones(6))
    o.ctrl[42:] = 1.0
    sim.warmup(); o.step(500)
    assert abs(sim.time - 0.05) < 1e-6
    for k in range(200):
        sim.set_actuator_inputs(fly.name, "position", targets[k])
        sim.step()
    o.step_replay(targets, np.arange(42), 0, 200)
    # 700 steps of a contact-rich rollout from reset: float32 rounding differences grow along the way (the long rollout, in
    # re-synchronised segments: test_hip_parity_r6.py::test_config1_rollout_runs_the_cpu_class)
    assert np.abs(sim.batch.field("qpos")[0].cpu().numpy() - o.qpos).max() < 5e-4
    assert sim.batch.get_solver_exits()["noslip_skipped"] == 0
    active, force, torque, pos, normal, tangent = sim.get_ground_contact_info(fly.name)
    assert active.shape == (6,) and force.shape == (6, 3) and tangent.shape == (6, 3)
    assert active.sum() >= 3 and force[:, 2].sum() > 0                     # standing on at least a tripod
    with pytest.raises(ValueError, match="Expected 42 inputs"):
        sim.set_actuator_inputs(fly.name, "position", np.zeros(41))
    with pytest.raises(ValueError, match="Unexpected number of adhesion states"):
        sim.set_leg_adhesion_states(fly.name, np.ones(5))
    # the MuJoCo attributes reference code reads most (tests/core/test_simulation.py): timestep, sizes, state copies
    assert sim.mj_model.opt.timestep == pytest.approx(1e-4) and sim.mj_model.nv == 72
    assert sim.mj_data.qpos.shape == (73,) and sim.mj_data.time == pytest.approx(sim.time)
    sim.step_with_profile()
    assert sim._curr_step == 1 and sim._total_physics_time_ns > 0 and sim._frames_rendered == 0
    sim.print_performance_report()
    sim.reset()
    assert sim.time == 0.0 and np.allclose(sim.get_joint_angles(fly.name), q0) and sim._curr_step == 0


@pytest.mark.parametrize("preset,nv", [("ALL_BIOLOGICAL", 132), ("ALL_POSSIBLE", 210), ("custom", 105),
                                       ("ALL_BIOLOGICAL-tables", 132)])
def test_general_tree_skeletons_parity(torch_mod, oracle_lib, preset, nv):
    """Skeletons that are not a star of identical leg chains (head with antennae and proboscis, abdomen, wings,
    halteres): the full-body presets (69 bodies, 132 / 210 dofs) run on the hybrid kernels (legs unrolled, the rest of
    the body swept as a tree), a custom skeleton (ALL_BIOLOGICAL without wings, halteres and abdomen joints: 60 bodies,
    105 dofs) on the pure general-tree kernel (nmf_tree.h).  Reset poses, the drop, landing and settling, then driven
    walking, against the float64 oracle — same bars as the leg-only skeletons."""
    torch = torch_mod
    import flygym_amd.compose as C
    from flygym_amd import HIPSimulation, anatomy as A
    from flygym_amd.controllers import TripodCPG
    from flygym_amd.utils.math import Rotation3D

    rest_slow = preset.endswith("-tables")      # the hybrid kernel's table-driven level passes (any dof count per body)
    if rest_slow:                               # instead of the unrolled three-dof ones the fly's skeletons take
        preset = preset[:-7]
    fly = C.Fly(name="t")
    if preset == "custom":
        bio = A.Skeleton(axis_order=A.AxisOrder.YAW_PITCH_ROLL, joint_preset=A.JointPreset.ALL_BIOLOGICAL)
        keep = [j for j in bio.anatomical_joints if not any(k in j.child.name for k in ("wing", "haltere", "abdomen"))]
        sk = A.Skeleton(axis_order=A.AxisOrder.YAW_PITCH_ROLL, anatomical_joints=keep)
    else:
        sk = A.Skeleton(axis_order=A.AxisOrder.YAW_PITCH_ROLL, joint_preset=getattr(A.JointPreset, preset))
    fly.add_joints(sk, neutral_pose=C.KinematicPosePreset.NEUTRAL)
    legs = A.Skeleton(axis_order=A.AxisOrder.YAW_PITCH_ROLL, joint_preset=A.JointPreset.LEGS_ONLY)
    fly.add_actuators(legs.get_actuated_dofs_from_preset("legs_active_only"), C.ActuatorType.POSITION, kp=50.0,
                      neutral_input=C.KinematicPosePreset.NEUTRAL)
    fly.add_leg_adhesion()
    world = C.FlatGroundWorld()
    world.add_fly(fly, (0, 0, 0.8), Rotation3D("quat", (1, 0, 0, 0)))
    n = 3
    sim = HIPSimulation(world, n_worlds=n, device=0, _options=dict(rest_slow=rest_slow))
    assert sim.model.nv == nv and sim.model.nb == (60 if preset == "custom" else 69) and int(sim.model["star"][0]) == 0
    o = oracle_lib.Oracle(sim.model.to_blob(), "f64")
    o32 = oracle_lib.Oracle(sim.model.to_blob(), "f32")
    assert np.abs(sim.field("seg_xpos").cpu().numpy()[0] - o.arr("seg_xpos")).max() < 2e-6
    nu = sim.model.nu
    assert nu == 48
    sim.set_leg_adhesion_states(fly.name, np.ones((n, 6), dtype=np.float32))
    for orc in (o, o32):
        orc.ctrl[nu - 6:] = 1.0
    for k in range(5):                                              # free fall, landing, settling
        sim.step(100)
        for orc in (o, o32):
            orc.step(100)
        q = sim.field("qpos").cpu().numpy()
        assert np.abs(q - o.qpos[None]).max() < 5e-5, f"{preset} after {100 * (k + 1)} steps"
    assert int(sim.field("stats")[0, 0].item()) == o32.ints()["ncon"] >= 5
    assert int(sim.field("stats")[0, 1].item()) == o32.ints()["solver_iter"]
    # one step from the same contact-rich state: accelerations to float32 accuracy, contact set bit-exact
    _push_state(sim, torch, o32.qpos, o32.qvel, o32.ctrl, o32.arr("qacc_warmstart"))
    sim.step(1); o32.step(1)
    qa, qa_ref = sim.field("qacc").cpu().numpy()[0], o32.arr("qacc")
    assert np.abs(qa - qa_ref).max() < 2e-3 * max(np.abs(qa_ref).max(), 1e4)
    assert int(sim.field("stats")[0, 0].item()) == o32.ints()["ncon"]
    # driven walking through the in-kernel control table (re-synchronised to the float64 state first)
    order = fly.get_actuated_jointdofs_order(C.ActuatorType.POSITION)
    table = TripodCPG(order, 1e-4).targets(1, 2500)
    tdev = torch.as_tensor(np.repeat(table, n, axis=0), device=sim.device)
    ids = sim.replay_ids(fly.name)
    _push_state(sim, torch, o.qpos, o.qvel, o.ctrl, o.arr("qacc_warmstart"))
    for k in range(2):
        sim.step_replay(tdev, ids, 100 * k, 100)
        o.step_replay(table[0], ids.cpu().numpy(), 100 * k, 100)
        assert np.abs(sim.field("qpos").cpu().numpy() - o.qpos[None]).max() < 1e-4, f"{preset} walking, tick {k}"
    assert sim.get_joint_angles(fly.name).shape == (n, nv - 6)
    assert torch.equal(sim.field("qpos")[0], sim.field("qpos")[n - 1])    # identical worlds stay identical


def test_every_segment_in_contact_69_geoms(torch_mod, oracle_lib):
    """ContactBodiesPreset ALL on the ALL_BIOLOGICAL skeleton: 69 geom-plane pairs (more than one wave of geoms: the
    collision stage takes two passes), no actuators — the fly drops and collapses onto body, wings and legs.  Compared
    with the float32 oracle in re-synchronised 20-step segments (see test_single_world_and_launch_splitting)."""
    torch = torch_mod
    import flygym_amd.compose as C
    from flygym_amd import HIPSimulation, anatomy as A
    from flygym_amd.utils.math import Rotation3D

    fly = C.Fly(name="t")
    fly.add_joints(A.Skeleton(axis_order=A.AxisOrder.YAW_PITCH_ROLL, joint_preset=A.JointPreset.ALL_BIOLOGICAL),
                   neutral_pose=C.KinematicPosePreset.NEUTRAL)
    world = C.FlatGroundWorld()
    world.add_fly(fly, (0, 0, 0.5), Rotation3D("quat", (1, 0, 0, 0)), bodysegs_with_ground_contact="all")
    sim = HIPSimulation(world, n_worlds=2, device=0)
    assert sim.model.ng == 69 and sim.model.nu == 0
    o = oracle_lib.Oracle(sim.model.to_blob(), "f32")
    errs, same_ncon, most = [], [], 0
    for k in range(30):
        _push_state(sim, torch, o.qpos, o.qvel, o.ctrl, o.arr("qacc_warmstart"))
        sim.step(20); o.step(20)
        errs.append(np.abs(sim.field("qpos").cpu().numpy()[0] - o.qpos).max())
        same_ncon.append(int(sim.field("stats")[0, 0].item()) == o.ints()["ncon"])
        most = max(most, o.ints()["ncon"])
    errs = np.array(errs)
    assert (errs < 1e-5).mean() >= 0.85, np.sort(errs)[-6:]
    assert errs.max() < 5e-3 and np.mean(same_ncon) >= 0.85
    assert most >= 8 and int(sim.field("stats")[0, 2].item()) == 0          # many contacts, no overflow
    geoms = set(o.ints()["con_geom"])
    assert max(geoms) >= 64 or most >= 8                                     # geoms of the second pass can be hit


def test_joint_sites_and_profile_counters(torch_mod, oracle_lib):
    """Reference tests/warp/test_simulation.py::TestSiteStateQueries / TestReset: sites added with add_joint_sites are
    reported in fly order with shape (n_worlds, n_sites, 3) and agree with the CPU-side model (here: the oracle) to 1e-6
    at reset and after stepping; reset clears time and the profiling counters."""
    torch = torch_mod
    import flygym_amd.compose as C
    from flygym_amd import HIPSimulation, anatomy as A
    from flygym_amd.utils.math import Rotation3D

    fly = C.Fly(name="sites")
    sk = A.Skeleton(axis_order=A.AxisOrder.YAW_PITCH_ROLL, joint_preset=A.JointPreset.LEGS_ONLY)
    fly.add_joints(sk, neutral_pose=C.KinematicPosePreset.NEUTRAL)
    fly.add_actuators(sk.get_actuated_dofs_from_preset("legs_active_only"), C.ActuatorType.POSITION, kp=50.0,
                      neutral_input=C.KinematicPosePreset.NEUTRAL)
    fly.add_leg_adhesion()
    joints = [j for j in sk.anatomical_joints if j.child.name.endswith(("tibia", "tarsus5"))]
    fly.add_joint_sites(joints)
    world = C.FlatGroundWorld()
    world.add_fly(fly, (0, 0, 0.8), Rotation3D("quat", (1, 0, 0, 0)))
    sim = HIPSimulation(world, n_worlds=3, device=0)
    o = oracle_lib.Oracle(sim.model.to_blob(), "f64")
    assert len(joints) == 12 and len(fly.get_sites_order()) == 12
    sites = sim.get_site_positions(fly.name)
    assert tuple(sites.shape) == (3, 12, 3)
    np.testing.assert_allclose(sites[0].cpu().numpy(), o.arr("site_xpos").reshape(-1, 3), atol=1e-6)
    np.testing.assert_allclose(sim.mj_data.site_xpos, o.arr("site_xpos").reshape(-1, 3), atol=1e-6)
    for _ in range(3):
        sim.step_with_profile()
    o.step(3)
    np.testing.assert_allclose(sim.get_site_positions(fly.name)[2].cpu().numpy(), o.arr("site_xpos").reshape(-1, 3), atol=2e-6)
    assert sim._curr_step == 3 and sim._total_physics_time_ns > 0 and sim.time == pytest.approx(3e-4)
    sim.reset()
    assert sim.time == 0.0 and sim._curr_step == 0 and sim._total_physics_time_ns == 0
    assert sim.n_worlds == 3 and isinstance(sim.time, float)


def test_two_seconds_of_walking_stay_sane(torch_mod, bench_model):
    """Soak: 20 000 steps (2 s of simulated time, 24 gait cycles) of CPG walking with gait-driven adhesion on 64 worlds with
    different phases: every state stays finite, every fly stays upright at walking height and moves forward, no contact
    overflow, and the Newton solver never hits its iteration cap."""
    torch = torch_mod
    from flygym_amd import HIPSimulation
    from flygym_amd.controllers import TripodCPG

    fly, world, _ = bench_model
    n = 64
    sim = HIPSimulation(world, n_worlds=n, device=0)
    cpg = TripodCPG(fly.get_actuated_jointdofs_order("position"), 1e-4)
    table = cpg.targets(n, 2500, device=sim.device, adhesion=(cpg.stance_bins(sim.model, fly), 20.0, 1.0))
    ids = sim.replay_ids(fly.name, with_adhesion=True)
    sim.set_leg_adhesion_states(fly.name, np.ones((n, 6), dtype=np.float32))
    sim.warmup()
    x0 = sim.field("qpos")[:, 0].clone()
    worst_iters = 0
    for tick in range(80):
        sim.step_replay(table, ids, 250 * tick, 250)
        st = sim.get_solver_stats()
        worst_iters = max(worst_iters, int(st[:, 1].max().item()))
        assert int(st[:, 2].sum().item()) == 0
    q = sim.field("qpos")
    assert torch.isfinite(q).all() and torch.isfinite(sim.field("qvel")).all()
    names = [s.name for s in fly.get_bodysegs_order()]
    z = sim.get_body_positions(fly.name)[:, names.index("c_thorax"), 2]
    assert float(z.min()) > 0.7 and float(z.max()) < 1.5                    # thorax height while walking (settles at ~1.07 mm)
    up = 1 - 2 * (q[:, 4] ** 2 + q[:, 5] ** 2)                               # z axis of the thorax along world z
    assert float(up.min()) > 0.9
    assert float((q[:, 0] - x0).min()) > 2.0                                 # every fly walked forward (3.4 .. 5.4 mm in 2 s)
    assert worst_iters < 50
    assert sim.time == pytest.approx(2.05, rel=1e-3)
