"""Round 6's reference surface on one MI355X: all of flygym's actuator types and a world with two flies.
Run from the repo root:  python examples/actuators_and_two_flies.py

`Fly.add_actuators(dofs, ActuatorType.X, **mjcf_attributes)` takes every member of the reference's `ActuatorType`
(`compose/fly.py:65-77`); several types may drive one joint.  `world.add_fly` takes several flies (`compose/world.py:95-149`):
the reference's flies never collide with each other, so each is stepped as its own batch behind the same per-fly calls.
"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
import torch

from flygym_amd import HIPSimulation
from flygym_amd.anatomy import ActuatedDOFPreset, AxisOrder, JointPreset, Skeleton
from flygym_amd.compose import ActuatorType, FlatGroundWorld, Fly, KinematicPosePreset
from flygym_amd.utils.math import Rotation3D

n = 256


def make_fly(name, preset):
    fly = Fly(name=name)
    fly.add_joints(Skeleton(axis_order=AxisOrder.YAW_PITCH_ROLL, joint_preset=preset), neutral_pose=KinematicPosePreset.NEUTRAL)
    return fly, fly.skeleton.get_actuated_dofs_from_preset(ActuatedDOFPreset.LEGS_ACTIVE_ONLY)


# fly 1: position servos on every active leg joint + a damper with a controllable damping scale on the same joints
alice, dofs = make_fly("alice", JointPreset.LEGS_ONLY)
alice.add_actuators(dofs, ActuatorType.POSITION, kp=50.0, neutral_input=KinematicPosePreset.NEUTRAL)
alice.add_actuators(dofs, ActuatorType.DAMPER, kv=2e-3, ctrlrange=(0.0, 5.0))
alice.add_leg_adhesion()

# fly 2: muscles (activation dynamics + force-length-velocity curves) on the femur-tibia pitch joints, integrated-velocity servos elsewhere
bob, dofs_b = make_fly("bob", JointPreset.LEGS_ACTIVE_ONLY)
knees = [d for d in dofs_b if "tibia" in d.name]
rest = [d for d in dofs_b if d not in knees]
bob.add_actuators(knees, ActuatorType.MUSCLE, lengthrange=(-3.0, 1.0), force=3.0, timeconst=(0.003, 0.01), forcerange=(-30.0, 30.0))
bob.add_actuators(rest, ActuatorType.INTVELOCITY, kp=40.0, kv=1e-3, actrange=(-2.5, 2.5))

world = FlatGroundWorld()
world.add_fly(alice, (0, 0, 0.8), Rotation3D("quat", (1, 0, 0, 0)))
world.add_fly(bob, (8.0, 0, 0.9), Rotation3D("quat", (0.9238795, 0, 0, 0.3826834)), add_ground_contact_sensors=False)

sim = HIPSimulation(world, n_worlds=n)              # one batch per fly behind the reference's per-fly API
sim.set_leg_adhesion_states("alice", np.ones((n, 6), dtype=np.float32))
sim.warmup(0.02)

g = torch.Generator(device=sim.device); g.manual_seed(0)
for tick in range(10):
    sim.set_actuator_inputs("alice", ActuatorType.DAMPER, torch.full((n, len(dofs)), float(tick % 5), device=sim.device))
    sim.set_actuator_inputs("bob", ActuatorType.MUSCLE, torch.rand((n, len(knees)), device=sim.device, generator=g))          # excitations in [0, 1]
    sim.set_actuator_inputs("bob", ActuatorType.INTVELOCITY, 4.0 * (torch.rand((n, len(rest)), device=sim.device, generator=g) - 0.5))   # set-point velocities, rad/s
    sim.step(20)
print(f"t = {sim.time * 1e3:.1f} ms")
print("alice: damper forces   max |f| =", float(sim.get_actuator_forces("alice", ActuatorType.DAMPER).abs().max()))
print("bob:   muscle forces   min / max =", float(sim.get_actuator_forces("bob", ActuatorType.MUSCLE).min()), float(sim.get_actuator_forces("bob", ActuatorType.MUSCLE).max()))
act = sim.for_fly("bob").field("act")                # the activation state (MuJoCo's act), one slot per actuator
print("bob:   activations     muscles", float(act[:, :len(knees)].mean()), " integrated set points |max|", float(act[:, len(knees):].abs().max()))
print("thorax x: alice", float(sim.get_body_positions("alice")[:, 0, 0].mean()), " bob", float(sim.get_body_positions("bob")[:, 0, 0].mean()))
print("batches:", {k: (v["kernel_family"], v["chunked"]) for k, v in sim.batch_info().items()})
assert torch.isfinite(sim.get_joint_angles("alice")).all() and torch.isfinite(sim.get_joint_angles("bob")).all()
