"""Ready-made model factories that mirror the reference's benchmark / test builders
(``src/flygym_demo/benchmark/time_gpu_simulation.py:21-64`` ``make_model`` and
``tests/warp/conftest.py:25-72`` ``build_gpu_sim``)."""

from __future__ import annotations

from .anatomy import ActuatedDOFPreset, AxisOrder, JointPreset, Skeleton
from .compose import ActuatorType, FlatGroundWorld, Fly, GeomFittingOption, KinematicPosePreset
from .utils.math import Rotation3D

__all__ = ["make_model"]


def make_model(
    joints_preset=JointPreset.LEGS_ONLY,
    actuated_dofs_preset=ActuatedDOFPreset.LEGS_ACTIVE_ONLY,
    actuator_type=ActuatorType.POSITION,
    position_gain=50.0,
    neutral_pose=KinematicPosePreset.NEUTRAL,
    spawn_position=(0, 0, 0.8),
    spawn_rotation=Rotation3D("quat", (1, 0, 0, 0)),
    simplify_geom=False,
    name="nmf",
):
    """Fly + flat-ground world with the reference benchmark's defaults; returns
    ``(fly, world, cam)`` exactly like the reference's ``make_model``."""
    option = GeomFittingOption.ALL_TO_CAPSULES if simplify_geom else GeomFittingOption.UNMODIFIED
    fly = Fly(name=name, geom_fitting_option=option)
    skeleton = Skeleton(axis_order=AxisOrder.YAW_PITCH_ROLL, joint_preset=joints_preset)
    fly.add_joints(skeleton, neutral_pose=neutral_pose)
    dofs = fly.skeleton.get_actuated_dofs_from_preset(actuated_dofs_preset)
    fly.add_actuators(dofs, actuator_type=actuator_type, kp=position_gain, neutral_input=neutral_pose)
    fly.add_leg_adhesion()
    fly.colorize()
    cam = fly.add_tracking_camera()
    world = FlatGroundWorld()
    world.add_fly(fly, spawn_position, spawn_rotation)
    return fly, world, cam
