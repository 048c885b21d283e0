"""The behaviours the reference's own simulation tests assert, on this repo's classes (MI355X).

Not a copy of those tests: a checklist of WHAT they establish, re-stated against ``HIPSimulation`` (the mirror of
``flygym.warp.GPUSimulation``; reference tests/warp/test_simulation.py, fixtures tests/warp/conftest.py:30-70) and
``Simulation`` (the mirror of ``flygym.Simulation``; reference tests/core/test_simulation.py, fixtures tests/conftest.py)
— same model construction calls, same method names and argument orders, torch tensors where the reference returns
``wp.array``.  Differences are asserted too, so that they are deliberate: rendering is handed off (``set_renderer`` raises),
``mj_model`` / ``mj_data`` are light stand-ins (no MuJoCo here).
"""

import warnings

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_mod():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    return torch


def _build(n_worlds=4, fly_name="warp_fly", joint_sites=False, world_cls="FlatGroundWorld", gpu=True):
    """The model the reference's fixtures build: LEGS_ONLY joints, position actuators (kp 50) on the LEGS_ACTIVE_ONLY dofs,
    leg adhesion, one tracking camera, spawned 0.8 mm over flat ground (or tethered)."""
    import flygym_amd.compose as C
    from flygym_amd import HIPSimulation, Simulation, anatomy as A
    from flygym_amd.utils.math import Rotation3D

    fly = C.Fly(name=fly_name)
    sk = A.Skeleton(axis_order=A.AxisOrder.YAW_PITCH_ROLL, joint_preset=A.JointPreset.LEGS_ONLY)
    pose = C.KinematicPosePreset.NEUTRAL.get_pose_by_axis_order(A.AxisOrder.YAW_PITCH_ROLL)
    fly.add_joints(sk, neutral_pose=pose)
    fly.add_actuators(sk.get_actuated_dofs_from_preset(A.ActuatedDOFPreset.LEGS_ACTIVE_ONLY), C.ActuatorType.POSITION, kp=50,
                      neutral_input=pose)
    if joint_sites:
        fly.add_joint_sites([A.AnatomicalJoint(A.BodySegment("c_thorax"), A.BodySegment("lf_coxa")),
                             A.AnatomicalJoint(A.BodySegment("c_thorax"), A.BodySegment("rf_coxa"))])
    fly.add_leg_adhesion()
    cam = fly.add_tracking_camera()
    world = getattr(C, world_cls)()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        world.add_fly(fly, [0, 0, 0.8], Rotation3D("quat", [1, 0, 0, 0]))
        sim = HIPSimulation(world, n_worlds=n_worlds, device=0) if gpu else Simulation(world)
    return sim, fly, cam, sk, pose


def test_gpu_class_construction_step_reset_and_profile(torch_mod, capsys):
    """reference tests/warp/test_simulation.py:43-160, 305-320, 350-367"""
    import flygym_amd.compose as C
    from flygym_amd import HIPSimulation, anatomy as A
    from flygym_amd.utils.math import Rotation3D

    sim, fly, cam, _, _ = _build(4)
    sim.reset()
    assert sim.n_worlds == 4 and _build(8, "multi_fly")[0].n_worlds == 8
    assert sim.time == pytest.approx(0.0) and isinstance(sim.time, float) and sim._curr_step == 0
    # noslip iterations are stripped, with a warning, as on the reference's batched path
    f2 = C.Fly(name="noslip_fly")
    f2.add_joints(A.Skeleton(axis_order=A.AxisOrder.YAW_PITCH_ROLL, joint_preset=A.JointPreset.LEGS_ONLY),
                  neutral_pose=C.KinematicPosePreset.NEUTRAL.get_pose_by_axis_order(A.AxisOrder.YAW_PITCH_ROLL))
    w2 = C.FlatGroundWorld()
    w2.noslip_iterations = 5
    w2.add_fly(f2, [0, 0, 0.8], Rotation3D("quat", [1, 0, 0, 0]))
    with pytest.warns(UserWarning, match="noslip"):
        HIPSimulation(w2, n_worlds=2, device=0)
    assert w2.noslip_iterations == 0
    # step and time
    dt = sim.mj_model.opt.timestep
    sim.step()
    assert sim.time == pytest.approx(dt, rel=1e-4)
    for _ in range(9):
        sim.step()
    assert sim.time == pytest.approx(10 * dt, rel=1e-3)
    # reset; profiled steps and their counters
    sim.reset()
    assert sim.time == pytest.approx(0.0)
    sim.step_with_profile()
    assert sim._curr_step == 1 and sim._total_physics_time_ns > 0
    sim.step_with_profile()
    sim.reset()
    assert sim._curr_step == 0 and sim._total_physics_time_ns == 0 and sim.time == pytest.approx(0.0)
    # warm-up
    sim.warmup(duration_s=0.001)
    assert sim.time > 0.0
    # performance report after profiled steps
    sim.reset()
    for _ in range(10):
        sim.step_with_profile()
    sim.print_performance_report()
    assert "PERFORMANCE" in capsys.readouterr().out
    # rendering is handed off: stated, not silently missing
    with pytest.raises(NotImplementedError):
        sim.set_renderer(cam, camera_res=(64, 64), worlds=[0, 1], use_gpu_batch_rendering=False)


def test_gpu_class_state_queries_and_control_inputs(torch_mod):
    """reference tests/warp/test_simulation.py:165-300"""
    torch = torch_mod
    from flygym_amd.compose import ActuatorType

    sim, fly, _, sk, _ = _build(4, "sim_sites_test_fly", joint_sites=True)
    sim.reset()
    ang, vel = sim.get_joint_angles(fly.name), sim.get_joint_velocities(fly.name)
    assert isinstance(ang, torch.Tensor) and ang.is_cuda and ang.ndim == 2 and ang.shape[0] == sim.n_worlds
    assert isinstance(vel, torch.Tensor) and vel.shape == ang.shape and ang.shape[1] == len(list(sk.iter_jointdofs()))
    bpos, brot = sim.get_body_positions(fly.name), sim.get_body_rotations(fly.name)
    assert bpos.shape[0] == brot.shape[0] == sim.n_worlds and bpos.shape[2] == 3 and brot.shape[2] == 4 and bpos.shape[1] == brot.shape[1]
    np.testing.assert_allclose(np.linalg.norm(brot.cpu().numpy(), axis=2), 1.0, atol=1e-5)
    # sites: one per added anatomical joint, world 0 equal to the CPU-side copy of world 0
    sim.step()
    spos = sim.get_site_positions(fly.name)
    assert isinstance(spos, torch.Tensor) and spos.shape == (sim.n_worlds, len(fly.get_sites_order()), 3) and spos.shape[1] == 2
    np.testing.assert_allclose(spos.cpu().numpy()[0], sim.mj_data.site_xpos, atol=1e-6)
    # control inputs: numpy and device arrays; one column per joint dof is accepted as the reference's own test passes it
    sim.reset()
    n_dofs = ang.shape[1]
    sim.set_actuator_inputs(fly.name, ActuatorType.POSITION, np.zeros((sim.n_worlds, n_dofs), dtype=np.float32))
    sim.set_actuator_inputs(fly.name, ActuatorType.POSITION, torch.zeros((sim.n_worlds, n_dofs), device=sim.device))
    sim.set_leg_adhesion_states(fly.name, np.ones((sim.n_worlds, 6), dtype=np.float32))
    sim.set_leg_adhesion_states(fly.name, torch.ones((sim.n_worlds, 6), device=sim.device))
    neutral = sim.get_joint_angles(fly.name).cpu().numpy().copy()
    for _ in range(50):
        sim.step()
    assert not np.allclose(neutral, sim.get_joint_angles(fly.name).cpu().numpy(), atol=1e-4)     # driven towards zero angles


def test_cpu_class_surface(torch_mod, capsys):
    """reference tests/core/test_simulation.py:31-400 on ``flygym_amd.Simulation`` (one world on the GPU behind the CPU
    class's unbatched numpy surface); tethered world as in the reference's ``simulation`` fixture."""
    import flygym_amd.compose as C
    from flygym_amd import Simulation
    from flygym_amd.compose import ActuatorType

    with pytest.raises(ValueError, match="at least one fly"):
        Simulation(C.TetheredWorld(name="emptyworld"))
    sim, fly, _, sk, pose = _build(fly_name="sim_fly", joint_sites=True, world_cls="TetheredWorld", gpu=False)
    sim.reset()
    assert sim.time == pytest.approx(0.0) and sim.mj_model is not None and sim.mj_data is not None
    t0 = sim.time
    sim.step()
    assert sim.time > t0
    for _ in range(4):
        sim.step()
    assert sim.time == pytest.approx(5 * sim.mj_model.opt.timestep, rel=1e-3)
    sim.reset()
    assert sim.time == pytest.approx(0.0)
    # joint angles: numpy, one per joint dof, the neutral pose at reset; velocities zero
    ang, vel = sim.get_joint_angles(fly.name), sim.get_joint_velocities(fly.name)
    dofs = list(sk.iter_jointdofs())
    assert isinstance(ang, np.ndarray) and isinstance(vel, np.ndarray) and len(ang) == len(vel) == len(dofs)
    lookup = pose.joint_angles_lookup_rad
    for i, d in enumerate(dofs):
        if d.name in lookup:
            assert ang[i] == pytest.approx(lookup[d.name], abs=0.2), d.name
    np.testing.assert_allclose(vel, 0.0, atol=1e-8)
    # bodies and sites
    bpos, brot = sim.get_body_positions(fly.name), sim.get_body_rotations(fly.name)
    assert bpos.ndim == 2 and bpos.shape[1] == 3 and brot.shape == (bpos.shape[0], 4) and bpos.shape[0] == len(fly.get_bodysegs_order())
    np.testing.assert_allclose(np.linalg.norm(brot, axis=1), 1.0, atol=1e-5)
    spos = sim.get_site_positions(fly.name)
    assert spos.shape == (2, 3) and [j.child.name for j in fly.get_sites_order()] == ["lf_coxa", "rf_coxa"]
    np.testing.assert_allclose(spos, sim.mj_data.site_xpos, atol=1e-6)
    # actuator io
    n_act = len(sk.get_actuated_dofs_from_preset("legs_active_only"))
    sim.set_actuator_inputs(fly.name, ActuatorType.POSITION, np.zeros(n_act))
    sim.step()
    forces = sim.get_actuator_forces(fly.name, ActuatorType.POSITION)
    assert isinstance(forces, np.ndarray) and len(forces) == n_act
    with pytest.raises(ValueError):
        sim.set_actuator_inputs(fly.name, ActuatorType.POSITION, np.zeros(n_act + 5))
    # adhesion: boolean arrays as the reference passes them
    sim.set_leg_adhesion_states(fly.name, np.ones(6, dtype=bool)); sim.step()
    sim.set_leg_adhesion_states(fly.name, np.zeros(6, dtype=bool)); sim.step()
    with pytest.raises(ValueError):
        sim.set_leg_adhesion_states(fly.name, np.ones(5, dtype=bool))
    # warm-up
    sim.reset()
    sim.warmup(duration_s=0.0)
    assert sim.time == pytest.approx(0.0)
    sim.warmup(duration_s=0.001)
    assert sim.time > 0.0
    # profiling counters and the report
    sim.reset()
    sim.step_with_profile()
    assert sim.time > 0.0 and sim._curr_step == 1 and sim._total_physics_time_ns > 0
    sim.reset()
    assert sim._curr_step == 0 and sim._total_physics_time_ns == 0
    for _ in range(5):
        sim.step_with_profile()
    sim.print_performance_report()
    assert "PERFORMANCE" in capsys.readouterr().out
    # ground contact info on flat ground: six legs
    flat, ffly, _, _, _ = _build(fly_name="test_fly", world_cls="FlatGroundWorld", gpu=False)
    flat.reset()
    active, forces, torques, positions, normals, tangents = flat.get_ground_contact_info(ffly.name)
    assert len(active) == 6 and all(a.shape == (6, 3) for a in (forces, torques, positions, normals, tangents))
