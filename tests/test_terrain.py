"""Terrain height maps (build-defined): the numpy specification and the oracle agree, and the
oracle's contacts sit on the local ground height."""

import numpy as np
import pytest

import flygym_amd.compose as C
from flygym_amd import anatomy as A
from flygym_amd.utils.math import Rotation3D


def _world(cls, **kw):
    fly = C.Fly(name="t")
    sk = A.Skeleton(axis_order=A.AxisOrder.YAW_PITCH_ROLL, joint_preset=A.JointPreset.LEGS_ONLY)
    fly.add_joints(sk, neutral_pose=C.KinematicPosePreset.NEUTRAL)
    fly.add_leg_adhesion()
    w = cls(**kw)
    w.add_fly(fly, (0.3, 0.2, 0.8), Rotation3D("quat", (1, 0, 0, 0)))
    return fly, w


def test_height_functions():
    _, g = _world(C.GappedTerrainWorld)
    np.testing.assert_array_equal(g.terrain_height([0.0, 0.99, 1.01, 1.29, 1.31, -0.1], 0.0), [0, 0, -2, -2, 0, -2])
    _, b = _world(C.BlocksTerrainWorld)
    np.testing.assert_array_equal(b.terrain_height([0.5, 1.5, 0.5, 1.5, -0.5], [0.5, 0.5, 1.5, 1.5, 0.5]), [0, 0.35, 0.35, 0, 0.35])
    _, m = _world(C.MixedTerrainWorld)
    assert m.terrain_height(1.0, 0.3) == 0.0                       # flat stripe
    assert m.terrain_height(4.0 + 1.1, 0.3) == -2.0                # gapped stripe, inside a gap
    assert m.terrain_height(8.0 + 1.5, 0.5) == 0.35                # blocks stripe, raised square
    assert m.compile_model()["terrain_params"][4] == 0.35 and int(m.compile_model()["terrain_type"][0]) == 3


@pytest.mark.parametrize("cls", [C.GappedTerrainWorld, C.BlocksTerrainWorld, C.MixedTerrainWorld])
def test_oracle_contacts_sit_on_the_terrain(cls, oracle_lib):
    fly, world = _world(cls)
    m = world.compile_model()
    o = oracle_lib.Oracle(m.to_blob(), "f64")
    o.ctrl[42:] = 1.0
    o.step(1200)
    i = o.ints()
    assert i["ncon"] >= 3 and np.abs(o.qvel).max() < 5.0
    pos = o.arr("con_pos").reshape(-1, 3)
    dist = o.arr("con_dist")
    h = world.terrain_height(pos[:, 0], pos[:, 1])
    # contact point = surface point - dist/2 along +z: it lies within |dist|/2 + margin of the local ground
    assert np.all(np.abs(pos[:, 2] - h) <= np.abs(dist) * 0.5 + 2e-3)
    # vertical balance at rest: the ground carries the weight plus whatever adhesion pulls are engaged (0..6)
    weight = m["body_mass"].sum() * 9810.0
    normal = o.arr("efc_force").sum()
    assert weight * 0.98 <= normal <= (weight + 6.0) * 1.02
