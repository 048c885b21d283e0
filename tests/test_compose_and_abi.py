"""Compose layer + compiled model structure (reference tests/core/test_compose.py cases that do
not need MuJoCo) and the C-ABI surface of libnmf_hip.so (no compute calls: no GPU here)."""

import ctypes

import numpy as np
import pytest

from flygym_amd import _native, anatomy as A
from flygym_amd.compiler.model import CompiledModel
from flygym_amd.compose import (ActuatorType, ContactParams, FlatGroundWorld, Fly, GeomFittingOption,
                                KinematicPosePreset)
from flygym_amd.utils.math import Rotation3D


def test_benchmark_model_dimensions(bench_model):
    fly, world, m = bench_model
    # nq 73 / nv 72 / nu 48 / 69 named bodies / 55 geom-plane pairs / 6 sensors x 16
    assert (m.nq, m.nv, m.nu, m.nseg, m.ng) == (73, 72, 48, 69, 55)
    assert m.nb == 49 and list(m["star"]) == [1, 6, 11, 8]
    assert len(fly.get_jointdofs_order()) == 66
    assert len(fly.get_actuated_jointdofs_order(ActuatorType.POSITION)) == 42
    assert int(m["n_sensor"][0]) == 6
    # all 6 tarsus5 geoms are capsules even with GeomFittingOption.UNMODIFIED (fly.py:585-589)
    caps = [fly.get_bodysegs_order()[s].name for s, t in zip(m["geom_seg"], m["geom_type"]) if t == 0]
    assert caps == [f"{leg}_tarsus5" for leg in A.LEGS]
    # total mass = rigging masses with the boundmass clamp + the massless attachment body's clamp
    assert m["body_mass"].sum() == pytest.approx(1.02531e-3, rel=1e-4)
    # keyframe: spawn pose then neutral angles; position actuators hold the neutral pose
    np.testing.assert_allclose(m["key_qpos"][:7], [0, 0, 0.8, 1, 0, 0, 0])
    neutral = [fly.jointdof_to_neutralangle[d] for d in fly.get_jointdofs_order()]
    np.testing.assert_allclose(m["key_qpos"][7:], neutral)
    assert np.count_nonzero(m["key_ctrl"][:42]) > 30 and not m["key_ctrl"][42:].any()


def test_reference_quirks_are_reproduced(bench_model):
    fly, world, m = bench_model
    # solimp passed as 4 numbers -> (0.98, 0.99, width 0.5, midpoint 3.0 clamped to 0.9999, power 2)
    np.testing.assert_allclose(m["pair_solimp"][0], [0.98, 0.99, 0.5, 0.9999, 2.0])
    np.testing.assert_allclose(m["pair_friction"][0], [1.0, 1.0, 0.02, 1e-4, 1e-4])
    np.testing.assert_allclose(m["pair_solref"][0], [2e-4, 1.0])
    assert m["pair_margin"][0] == 1e-3
    # adhesion ctrlrange (1, 100), limited (fly.py:434-440)
    adh = m["act_type"] == 1
    np.testing.assert_array_equal(m["act_ctrlrange"][adh], np.tile([1.0, 100.0], (6, 1)))
    # right-side roll / yaw axes are negated, pitch is not (fly.py:279-283)
    names = [d.name for d in fly.get_jointdofs_order()]
    ax = m["dof_axis"][6:]
    assert tuple(ax[names.index("c_thorax-lf_coxa-yaw")]) == (1, 0, 0)
    assert tuple(ax[names.index("c_thorax-rf_coxa-yaw")]) == (-1, 0, 0)
    assert tuple(ax[names.index("c_thorax-rf_coxa-pitch")]) == (0, 1, 0)
    assert tuple(ax[names.index("rf_coxa-rf_trochanterfemur-roll")]) == (0, 0, -1)
    # boundmass raised the tiny tarsal masses
    assert m["body_mass"].min() == pytest.approx(1e-6)


def test_blob_roundtrip(bench_model):
    _, _, m = bench_model
    blob = m.to_blob()
    back = CompiledModel.from_blob(blob)
    assert back.keys() == m.keys()
    for k in m:
        np.testing.assert_array_equal(back[k], m[k])
    with pytest.raises(ValueError):
        CompiledModel.from_blob(b"garbage" * 10)


def test_compose_errors_and_variants():
    fly = Fly(name="f2", geom_fitting_option=GeomFittingOption.ALL_TO_CAPSULES)
    sk = A.Skeleton(axis_order=A.AxisOrder.YAW_PITCH_ROLL, joint_preset=A.JointPreset.LEGS_ACTIVE_ONLY)
    fly.add_joints(sk, neutral_pose=KinematicPosePreset.NEUTRAL)
    fly.add_leg_adhesion()
    with pytest.raises(ValueError):
        fly.add_leg_adhesion()
    world = FlatGroundWorld()
    with pytest.raises(ValueError):
        world.add_fly(fly, (0, 0, 1), Rotation3D("euler", (0, 0, 0)))
    world = FlatGroundWorld()
    world.add_fly(fly, (0, 0, 1), Rotation3D("quat", (1, 0, 0, 0)), ground_contact_params=ContactParams(sliding_friction=2.0))
    m = world.compile_model()
    assert m.nv == 48 and m.nu == 6 and list(m["star"]) == [1, 6, 7, 4]
    assert (m["geom_type"] == 0).all()            # every collision geom is a capsule
    assert m["pair_friction"][0][0] == 2.0
    with pytest.raises(ValueError):
        world.add_fly(fly, (0, 0, 1), Rotation3D("quat", (1, 0, 0, 0)))


def test_actuator_types(oracle_lib):
    """Reference ``ActuatorType`` (compose/fly.py:65-77): the stateless affine servos are compiled — position (gain kp, bias
    (-kp, -kv)), velocity (gain kv, bias (0, -kv)), motor — next to adhesion; several types may drive one joint
    (compose/fly.py:310-312): the first actuator of a dof runs in the affine pass, later ones are rows of ``act_general`` with the
    same law (round 6; the stateful types and damper: tests/test_oracle_actuator_types.py).
    A velocity-actuated leg joint in the oracle reports kv (ctrl - qd), clamped to the force range."""
    from flygym_amd.compose import ActuatorType

    fly = Fly(name="v")
    sk = A.Skeleton(axis_order=A.AxisOrder.YAW_PITCH_ROLL, joint_preset=A.JointPreset.LEGS_ONLY)
    fly.add_joints(sk, neutral_pose=KinematicPosePreset.NEUTRAL)
    dofs = sk.get_actuated_dofs_from_preset("legs_active_only")
    fly.add_actuators(dofs, ActuatorType.POSITION, kp=50.0, kv=0.5, neutral_input=KinematicPosePreset.NEUTRAL)
    fly.add_actuators(dofs[:3], "velocity", kv=2.0, forcerange=(-4.0, 4.0))
    fly.add_actuators(dofs[3:5], ActuatorType.MOTOR)
    world = FlatGroundWorld()
    world.add_fly(fly, (0, 0, 0.8), Rotation3D("quat", (1, 0, 0, 0)))
    m = world.compile_model()
    n = len(dofs)
    assert m.nu == n + 5
    np.testing.assert_array_equal(m["act_gain"][:n], 50.0)
    np.testing.assert_array_equal(m["act_bias"][:n], np.tile([-50.0, -0.5], (n, 1)))
    # the velocity and motor actuators sit on dofs the position actuators already drive: gain 0 in the affine pass, their law in
    # act_general (flags, no dynamics, fixed gain, affine bias)
    np.testing.assert_array_equal(m["act_gain"][n:], 0.0)
    np.testing.assert_array_equal(m["act_bias"][n:], 0.0)
    np.testing.assert_array_equal(m["act_limited"][n:, 0], 0)
    g = m["act_general"]
    assert g.shape == (m.nu, 32) and not g[:n].any()
    np.testing.assert_array_equal(g[n:n + 3, 0], 3.0 + 256.0 * m["act_trn"][n:n + 3])          # on | forcelimited | dof << 8
    np.testing.assert_array_equal(g[n:n + 3, 1:4], np.tile([0.0, 0.0, 1.0], (3, 1)))
    np.testing.assert_array_equal(g[n:n + 3, 9], 2.0)
    np.testing.assert_array_equal(g[n:n + 3, 18:21], np.tile([0.0, 0.0, -2.0], (3, 1)))
    np.testing.assert_array_equal(g[n + 3:, 9], 1.0)
    np.testing.assert_array_equal(g[n + 3:, 18:21], 0.0)
    assert [d.name for d in fly.get_actuated_jointdofs_order(ActuatorType.VELOCITY)] == [d.name for d in dofs[:3]]
    o = oracle_lib.Oracle(m.to_blob(), "f64")
    o.ctrl[n:n + 3] = [1.0, -0.5, 10.0]
    o.qvel[6 + np.array(m["act_trn"][n:n + 3]) - 6] = [0.25, 0.0, 0.0]
    qd = o.qvel[np.array(m["act_trn"][n:n + 3])].copy()
    o.forward()
    np.testing.assert_allclose(o.arr("actuator_force")[n:n + 3], np.clip(2.0 * (np.array([1.0, -0.5, 10.0]) - qd), -4.0, 4.0), rtol=1e-12)


def test_abi_exports_every_declared_symbol():
    _native.build()
    lib = ctypes.CDLL(str(_native.LIB_PATH))
    declared = _native.exported_symbols()
    assert len(declared) >= 14
    for name in declared:
        assert hasattr(lib, name), f"libnmf_hip.so does not export {name}"


def test_abi_struct_layout_and_plan_size():
    """The ctypes mirror of nmf_eye_params has the layout the library was compiled with; plan buffer sizing."""
    from flygym_amd.vision import _EyeParams

    L = _native.lib()
    assert ctypes.sizeof(_EyeParams) == L.nmf_eye_params_size()
    n_pix = 512 * 450
    assert L.nmf_retina_plan_bytes(n_pix) >= (n_pix // 16) * 20 + 4 and L.nmf_retina_plan_bytes(n_pix) % 16 == 0
    assert L.nmf_retina_plan_bytes(17) == 0                 # not a multiple of 16 pixels: no plan


def test_abi_model_parse_and_loud_failure_without_gpu(bench_model):
    import torch

    _, _, m = bench_model
    L = _native.lib()
    blob = m.to_blob()
    h = L.nmf_model_create(blob, len(blob))
    assert h
    dims = (ctypes.c_int32 * 10)()
    assert L.nmf_model_dims(h, dims) == 0
    assert list(dims)[:7] == [73, 72, 48, 49, 69, 55, 0] and dims[9] == 1
    # the most contacts the contact set can make at once: 6 capsules x 2 end spheres + 49 hulls x 4 vertices = 208
    gt = np.asarray(m["geom_type"]).ravel()
    assert L.nmf_model_contact_bound(h) == int(2 * (gt == 0).sum() + 4 * (gt == 1).sum()) > dims[7] == 48
    assert L.nmf_batch_set_contact_capacity(None, 10) < 0 and b"null batch" in L.nmf_last_error()
    assert not L.nmf_model_create(b"NOTAMODEL" * 4, 36)
    assert b"NMFMODEL" in L.nmf_last_error()
    if not torch.cuda.is_available():
        assert not L.nmf_batch_create(h, 4, 0)            # no silent CPU fallback
        assert b"hipSetDevice" in L.nmf_last_error()
        from flygym_amd import HIPSimulation, make_model

        with pytest.raises(_native.NativeError):
            HIPSimulation(make_model()[1], n_worlds=2)
    L.nmf_model_destroy(h)


def test_product_never_imports_the_oracle():
    import re
    from pathlib import Path

    pkg = Path(_native.__file__).parent
    for f in list(pkg.rglob("*.py")) + list(pkg.rglob("*.hip")) + list(pkg.rglob("*.h")):
        text = f.read_text()
        assert not re.search(r"^\s*(import|from)\s+oracle\b", text, re.M), f
        assert "nmf_oracle" not in text or "oracle/nmf_oracle.c" in text, f   # comments may cite it


def test_reference_utility_surface(bench_model, capsys):
    """Helpers user code imports from the reference's ``flygym.utils`` / ``flygym.compose`` (``utils/math.py``,
    ``utils/exceptions.py``, ``utils/profiling.py``, ``utils/pose_conversion.py``, ``compose/base.py::compile``)."""
    import flygym_amd
    from flygym_amd.anatomy import AxisOrder, JointPreset, Skeleton
    from flygym_amd.compose import KinematicPose, KinematicPosePreset
    from flygym_amd.utils.exceptions import FlyGymInternalError
    from flygym_amd.utils.math import Tree, orderedset
    from flygym_amd.utils.pose_conversion import get_body_names, get_xpos0_xquat0, qpos_to_kinematic_pose
    from flygym_amd.utils.profiling import print_perf_report, print_perf_report_parallel

    assert orderedset([3, 1, 3, 2, 1]) == [3, 1, 2]
    tree = Tree(["a", "b", "c", "d"], [("a", "b"), ("a", "c"), ("c", "d")])
    assert list(tree.dfs_edges("a")) == [("a", "b"), ("a", "c"), ("c", "d")]
    assert list(tree.dfs_edges("d"))[0] == ("d", "c")
    for nodes, edges in (([1, 2, 3], [(1, 2), (2, 3), (3, 1)]), ([1, 2, 3, 4], [(1, 2), (3, 4)]), ([1, 2], [(1, 2), (1, 2)]),
                         ([1, 2], [(1, 2), (1, 1)]), ([1, 1, 2], [(1, 2)]), ([1, 2], [(1, 99)])):
        with pytest.raises(ValueError):
            Tree(nodes, edges)
    with pytest.raises(ValueError):
        list(tree.dfs_edges("zz"))
    assert issubclass(FlyGymInternalError, Exception)
    sk = Skeleton(axis_order=AxisOrder.YAW_PITCH_ROLL, joint_preset=JointPreset.ALL_BIOLOGICAL)
    assert len(list(sk.get_tree().dfs_edges(sk.body_segments[0]))) == len(sk.body_segments) - 1
    # pose files on disk, as the reference's tutorials address them
    path = flygym_amd.assets_dir / "model/pose/neutral/pitch_roll_yaw.yaml"
    assert path.is_file() and KinematicPosePreset.NEUTRAL.get_dir() == path.parent
    pose = KinematicPose(path=path)
    assert pose.axis_order is AxisOrder.PITCH_ROLL_YAW and len(pose.joint_angles_lookup_rad) > 60
    with pytest.raises(ValueError, match="axis_order"):
        KinematicPose(path=path, axis_order=AxisOrder.PITCH_ROLL_YAW)
    # compile() -> (model, data) for worlds and standalone flies
    fly, world, m = bench_model
    model, data = world.compile()
    assert model is m and data.qpos.shape == (73,) and data.ctrl.shape == (48,) and data.time == 0.0
    assert (model.nq, model.nv, model.nu, model.nbody, model.njnt) == (73, 72, 48, 70, 67)
    from flygym_amd.compose import Fly

    bare = Fly(name="bare")
    bare.add_joints(Skeleton(axis_order=AxisOrder.YAW_PITCH_ROLL, joint_preset=JointPreset.LEGS_ONLY),
                    neutral_pose=KinematicPosePreset.NEUTRAL)
    fm, fd = bare.compile()
    assert fm.nv == fm.nq == 66 and fd.qpos.shape == (66,) and fm.compiled.nv == 72
    names = get_body_names(fm)
    assert names[0] == "world" and len(names) == fm.nbody == 70
    xpos, xquat = get_xpos0_xquat0(fm, fd)
    assert xpos.shape == (70, 3) and np.allclose(np.linalg.norm(xquat[1:], axis=1), 1.0, atol=1e-9)
    back = qpos_to_kinematic_pose(fm, fd.qpos, AxisOrder.YAW_PITCH_ROLL)
    ref = KinematicPosePreset.NEUTRAL.get_pose_by_axis_order(AxisOrder.YAW_PITCH_ROLL)
    key = "c_thorax-rf_coxa-roll"
    assert back.joint_angles_lookup_rad[key] == pytest.approx(ref.joint_angles_lookup_rad[key])      # mirrored from the left
    # performance reports
    print_perf_report(1_000_000, 500_000, 100, 10, 1e-3)
    print_perf_report_parallel(1_000_000, 0, 100, 0, 1e-3, 8, 0)
    out = capsys.readouterr().out
    assert "PERFORMANCE" in out and "Physics" in out and "No frames" in out and "parallelized" in out.lower()
    with pytest.raises(ValueError, match="n_steps"):
        print_perf_report(1, 0, 0, 0, 1e-3)


def test_semantics_changed_after_compile_recompile_the_world(bench_model):
    """world.semantics is a mutable object read at compile time: a flag flipped after the first compile must reach the
    next compile_model() instead of being shadowed by the cached model."""
    from flygym_amd.models import make_model

    fly, world, _ = make_model()
    a = world.compile_model()
    assert world.compile_model() is a                                   # cached while nothing changes
    world.semantics.pyramid_R = "plain"
    b = world.compile_model()
    assert b is not a and int(b["sem_options"][0]) == 1 and int(a["sem_options"][0]) == 0
    world.semantics.mesh_inertia = "convex"                              # a compile-time semantic: the inertias change
    c = world.compile_model()
    assert c is not b and not np.allclose(c["body_inertia"], b["body_inertia"])
    world.semantics.pyramid_R = "nonsense"
    with pytest.raises(ValueError):
        world.compile_model()


def test_replay_device_path_falls_back_for_clips_the_kernel_cannot_hold():
    """MotionSnippet.get_joint_angles_device: clips outside the kernel's 6..1536 frames take the scipy path."""
    torch = pytest.importorskip("torch")
    from flygym_amd.anatomy import AxisOrder, JointPreset, Skeleton
    from flygym_amd.replay import MotionSnippet

    ms = MotionSnippet()
    sk = Skeleton(axis_order=AxisOrder.YAW_PITCH_ROLL, joint_preset=JointPreset.LEGS_ONLY)
    order = sk.get_actuated_dofs_from_preset("legs_active_only")
    ms.joint_angles = np.tile(ms.joint_angles, (3, 1, 1))                # 1980 frames: more than the kernel's LDS column
    got = ms.get_joint_angles_device(1e-3, order, "cpu")
    want = ms.get_joint_angles(1e-3, order).astype(np.float32)
    assert tuple(got.shape) == want.shape and got.dtype == torch.float32
    np.testing.assert_array_equal(got.numpy(), want)


def test_a_world_takes_several_flies():
    """Reference ``BaseWorld.add_fly`` (compose/world.py:95-149): several flies per world, unique names; each fly keeps its own
    spawn pose, contact set and sensors, and compiles to its own model (the reference's flies never collide with each other:
    compose/fly.py:609-610, compose/world.py:300-309)."""
    def mk(name, preset):
        f = Fly(name=name)
        f.add_joints(A.Skeleton(axis_order=A.AxisOrder.YAW_PITCH_ROLL, joint_preset=preset), neutral_pose=KinematicPosePreset.NEUTRAL)
        f.add_actuators(f.skeleton.get_actuated_dofs_from_preset("legs_active_only"), "position", kp=50.0)
        return f

    w = FlatGroundWorld()
    w.add_fly(mk("a", A.JointPreset.LEGS_ONLY), (0, 0, 0.8), Rotation3D("quat", (1, 0, 0, 0)))
    w.add_fly(mk("b", A.JointPreset.LEGS_ACTIVE_ONLY), (5, 0, 1.0), Rotation3D("quat", (1, 0, 0, 0)), add_ground_contact_sensors=False,
              ground_contact_params=ContactParams(sliding_friction=2.0))
    with pytest.raises(ValueError, match="already exists"):
        w.add_fly(mk("a", A.JointPreset.LEGS_ONLY), (0, 0, 1), Rotation3D("quat", (1, 0, 0, 0)))
    assert list(w.fly_lookup) == ["a", "b"] and set(w.world_dof_neutral_states) == {"a/", "b/"}
    with pytest.raises(ValueError, match="holds 2 flies"):
        w.compile_model()
    ma, mb = w.compile_model("a"), w.compile_model("b")
    assert (ma.nv, mb.nv) == (72, 48)
    np.testing.assert_allclose(ma["key_qpos"][:3], [0, 0, 0.8]); np.testing.assert_allclose(mb["key_qpos"][:3], [5, 0, 1.0])
    assert int(ma["n_sensor"][0]) == 6 and int(mb["n_sensor"][0]) == 0
    assert ma["pair_friction"][0][0] == 1.0 and mb["pair_friction"][0][0] == 2.0
    # the view of one fly is the single-fly world: the same model as a world built with that fly alone
    alone = FlatGroundWorld()
    alone.add_fly(mk("a", A.JointPreset.LEGS_ONLY), (0, 0, 0.8), Rotation3D("quat", (1, 0, 0, 0)))
    assert alone.compile_model().digest() == ma.digest()
    with pytest.raises(KeyError):
        w.compile_model("c")


def test_per_joint_parameters_and_global_options_reach_the_model(oracle_lib):
    """What the reference's tutorial 1bis does through ``fly.mjcf_root`` (dm_control): a different stiffness for the tarsal joints
    only, another timestep.  Here: ``Fly.set_joint_params`` / ``fly.joint_params`` and ``fly.mujoco_globals`` — the compiled model
    carries them and the oracle's passive force follows."""
    fly = Fly(name="t")
    sk = A.Skeleton(axis_order=A.AxisOrder.YAW_PITCH_ROLL, joint_preset=A.JointPreset.LEGS_ONLY)
    fly.add_joints(sk, neutral_pose=KinematicPosePreset.NEUTRAL)
    tarsal = [d for d in fly.get_jointdofs_order() if d.child.link.startswith("tarsus") and d.child.link != "tarsus1"]
    assert len(tarsal) == 24
    fly.set_joint_params(tarsal, stiffness=5.0, damping=0.25)
    with pytest.raises(ValueError, match="negative"):
        fly.set_joint_params(tarsal[:1], damping=-1.0)
    fly.mujoco_globals["option"]["timestep"] = 2e-4
    world = FlatGroundWorld()
    world.add_fly(fly, (0, 0, 0.8), Rotation3D("quat", (1, 0, 0, 0)))
    m = world.compile_model()
    order = fly.get_jointdofs_order()
    is_t = np.array([d in tarsal for d in order])
    np.testing.assert_array_equal(m["dof_stiffness"][6:][is_t], 5.0)
    np.testing.assert_array_equal(m["dof_stiffness"][6:][~is_t], 10.0)
    np.testing.assert_array_equal(m["dof_damping"][6:][is_t], 0.25)
    assert float(m["opt_timestep"][0]) == 2e-4
    o = oracle_lib.Oracle(m.to_blob(), "f64")
    o.qpos[7:] += 0.1                      # every hinge 0.1 rad off its spring reference
    o.forward()
    qp = o.arr("qfrc_passive")[6:]
    np.testing.assert_allclose(qp[is_t], -0.5, rtol=1e-12); np.testing.assert_allclose(qp[~is_t], -1.0, rtol=1e-12)
    o.step(1)
    assert o.arr("time")[0] == pytest.approx(2e-4)
