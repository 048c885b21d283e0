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
    nrm = o.arr("con_frame").reshape(-1, 9)[:, :3]
    top = nrm[:, 2] == 1.0
    h = world.terrain_height(pos[:, 0], pos[:, 1])
    # a contact with the top of a cell: surface point - dist/2 along +z lies within |dist|/2 + margin of the local ground
    assert np.all(np.abs(pos[top, 2] - h[top]) <= np.abs(dist[top]) * 0.5 + 2e-3)
    # a contact with a side face (round 3: the cells are boxes): horizontal unit normal, the point lies on the face —
    # stepping half a millimetre against the normal lands inside a cell that is higher than the contact point
    for p_c, n_c, d_c in zip(pos[~top], nrm[~top], dist[~top]):
        assert n_c[2] == 0.0 and abs(np.abs(n_c).sum() - 1.0) < 1e-12
        inside = p_c - (abs(d_c) * 0.5 + 5e-3) * n_c
        assert world.terrain_height(inside[0], inside[1]) > p_c[2]
        assert world.terrain_height(*(p_c + 5e-3 * n_c)[:2]) <= p_c[2] + 2e-3
    # vertical balance at rest: the ground carries the weight plus whatever adhesion pulls are engaged (0..6); a leg
    # braced against a side face may carry part of it by friction
    weight = m["body_mass"].sum() * 9810.0
    f = o.arr("efc_force").reshape(-1, 4)
    fr = o.arr("con_frame").reshape(-1, 3, 3)
    mu = 1.0
    Fc = f.sum(1)[:, None] * fr[:, 0] + (mu * (f[:, 0] - f[:, 1]))[:, None] * fr[:, 1] + (mu * (f[:, 2] - f[:, 3]))[:, None] * fr[:, 2]
    assert weight * 0.98 <= Fc[:, 2].sum() <= (weight + 6.0) * 1.02
    assert np.abs(Fc[:, :2].sum(0)).max() < 0.2 * weight             # nearly settled: little net horizontal push


TERRAINS = {"gapped": (1, (1.0, 0.3, 2.0, 0.0)), "blocks": (2, (1.3, 0.35, 0.0, 0.0)), "mixed": (3, (1.3, 0.3, 2.0, 4.0))}


def test_terrain_probe_known_answers():
    """The box-terrain probe rule (compose.world.terrain_probe) on cases worked by hand."""
    from flygym_amd.compose.world import terrain_probe

    inf = np.inf
    g = TERRAINS["gapped"]
    assert terrain_probe(*g, (0.5, 0.0, 0.1)) == (0.1, inf, 0)                                # over a block, free of walls
    d = terrain_probe(*g, (1.001, 0.0, -0.5))                                                 # in the gap, 1 um off the block to its left
    assert d[0] == 1.5 and d[1] == pytest.approx(1e-3) and d[2] == 1                          # that face looks along +x
    d = terrain_probe(*g, (1.299, 0.0, -0.5))
    assert d[1] == pytest.approx(1e-3) and d[2] == 2                                          # the next block's face looks along -x
    d = terrain_probe(*g, (0.998, 0.0, -0.5))                                                 # 2 um INSIDE the block, 0.5 mm below its top
    assert d[0] == inf and d[1] == pytest.approx(-2e-3) and d[2] == 1                         # out through the face, not the top
    assert terrain_probe(*g, (0.5, 0.0, -0.001)) == (-0.001, inf, 0)                          # just under the top, mid-block: up
    d = terrain_probe(*g, (1.15, 0.0, -1.9), 0.2)                                             # a sphere wider than the gap's half width
    assert d[0] == pytest.approx(-0.1) and d[1] == pytest.approx(-0.05) and d[2] in (1, 2)    # both faces 50 um inside it
    b = TERRAINS["blocks"]
    d = terrain_probe(*b, (1.29, 0.5, 0.1), 0.05)                                             # low cell, raised neighbour at x > 1.3
    assert d[0] == pytest.approx(0.05) and d[1] == pytest.approx(-0.04) and d[2] == 2
    assert terrain_probe(*b, (1.29, 0.5, 0.45), 0.05)[1:] == (inf, 0)                         # above the neighbour's top: no face
    # a sphere rolling off the raised cell [1.3, 2.6) x [0, 1.3): over the edge its top still carries it, further out its face
    assert terrain_probe(*b, (2.599, 0.5, 0.379), 0.03) == (pytest.approx(-0.001), inf, 0)
    d = terrain_probe(*b, (2.6001, 0.5, 0.379), 0.03)                                         # centre 29 um above the edge, 0.1 um out
    assert d[0] == pytest.approx(-0.001) and d[1] == pytest.approx(0.47) and d[2] == 3        # (the only face in sight: the raised cell at y < 0)
    d = terrain_probe(*b, (2.6295, 0.5, 0.379), 0.03)                                         # 29.5 um out: nearer the face than the top
    assert d[0] == pytest.approx(0.349) and d[1] == pytest.approx(-0.0005) and d[2] == 1
    d = terrain_probe(*b, (1.0, 1.299, 0.2))                                                  # y faces: raised cell at y > 1.3
    assert d[1] == pytest.approx(1e-3) and d[2] == 4
    m = TERRAINS["mixed"]
    d = terrain_probe(*m, (2.0, 0.3, 0.05))                                                    # flat stripe, 2 mm from its ends
    assert d[0] == 0.05 and d[1] >= 2.0 - 1e-12
    d = terrain_probe(*m, (7.999, 0.65, 0.2))      # end of the gapped stripe (x < 8), the blocks stripe's first square is raised?
    h_next = float(np.asarray(__import__("flygym_amd.compose.world", fromlist=["x"])._terrain_height(3, m[1], np.float64(8.0001), np.float64(0.65))))
    assert (d[2] == 2 and d[1] == pytest.approx(1e-3)) == (h_next > 0.2)


@pytest.mark.parametrize("name", ["gapped", "blocks", "mixed"])
def test_probe_depth_is_continuous_across_cell_edges(name):
    """A sphere carried along straight lines over the terrain (above, at and below the tops; along x, y and a diagonal):
    the deepest penetration the rule reports moves by no more than the sphere does — no jump when its centre crosses a
    cell's edge (round 3's rule jumped by about rho there).  Lines that pass INSIDE a box are left out: there the rule's
    way out changes from one face to another by design."""
    from flygym_amd.compose.world import terrain_probe

    t = TERRAINS[name]
    rho, step = 0.03, 5e-4
    tops = {"gapped": (0.0,), "blocks": (0.0, 0.35), "mixed": (0.0, 0.35)}[name]
    worst = 0.0
    for top in tops:
        for dz in (-0.004, -0.001, 0.0, 0.01, 0.029):                # lowest point from 4 um inside a top to just below the centre's level
            z = top + rho + dz
            for (x0, y0, ux, uy) in ((-1.0, 0.5, 1.0, 0.0), (0.65, -1.0, 0.0, 1.0), (-1.0, -0.9, 0.8, 0.6), (3.5, 0.2, 1.0, 0.0)):
                prev = None
                for k in range(int(6.0 / step)):
                    x, y = x0 + ux * k * step, y0 + uy * k * step
                    h_here = float(np.asarray(__import__("flygym_amd.compose.world", fromlist=["x"])._terrain_height(t[0], t[1], np.float64(x), np.float64(y))))
                    if z - rho < h_here - 0.005:                      # the line runs inside this box: not this test's subject
                        prev = None
                        continue
                    dtop, dwall, _ = terrain_probe(*t, (x, y, z), rho)
                    depth = min(dtop, dwall, 0.0)                  # penetration only: clear of everything reads 0
                    if prev is not None:
                        worst = max(worst, abs(depth - prev))
                        assert abs(depth - prev) <= step * 1.0001 + 1e-12, (name, x, y, z, prev, depth)
                    prev = depth
    assert worst > 0.0


@pytest.mark.parametrize("name", ["gapped", "blocks", "mixed"])
@pytest.mark.parametrize("precision", ["f64", "f32"])
def test_oracle_contacts_follow_the_probe_rule(oracle_lib, name, precision):
    """The oracle's collision of a sphere with a box terrain against the numpy rule, at 400 seeded positions biased
    towards cell boundaries (both sides of them, above and below the tops)."""
    from flygym_amd.compose.world import terrain_probe
    from tiny_models import sphere_on_plane

    t = TERRAINS[name]
    rho, margin = 0.05, 1e-3
    m = sphere_on_plane(radius=rho, margin=margin, terrain=t, start_xy=(0.0, 0.0))
    o = oracle_lib.Oracle(m.to_blob(), precision)
    rng = np.random.default_rng(3)
    n_wall = n_top = 0
    tol = 1e-9 if precision == "f64" else 2e-5
    for k in range(400):
        edge = rng.choice([0.0, 1.0, 1.3, 2.6, 4.0, 5.0, 5.3, 8.0, 9.3, -1.3, -2.6])
        x = edge + rng.choice([-1, 1]) * rng.choice([rho + 5e-4, rho - 2e-3, 3e-3, 0.2, 0.6]) if k % 4 else rng.uniform(-6, 12)
        y = rng.choice([0.0, 1.3, -1.3, 2.6]) + rng.choice([-1, 1]) * rng.choice([rho + 5e-4, 4e-3, 0.3]) if k % 3 == 0 else rng.uniform(-3, 3)
        z = rng.choice([rho + 5e-4, rho - 1e-3, 0.2, 0.35 + rho + 5e-4, -0.5, -1.9 + rho])
        # keep clear of exact ties of the rule (a boundary within float rounding of the probe)
        o.qpos[:3] = (x, y, z)
        o.forward()
        dtop, dwall, wall = terrain_probe(*t, (x, y, z), rho)
        want = []
        if dtop <= margin:
            want.append((dtop, (0.0, 0.0, 1.0)))
        if wall and dwall <= margin:
            want.append((dwall, ((1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0))[wall - 1]))
        if any(abs(d - margin) < 1e-5 for d, _ in want) or abs(dtop - margin) < 1e-5 or abs(dwall - margin) < 1e-5:
            continue                                             # the margin test itself would be a tie in float32
        def outcome(q):
            a_, b_, c_ = terrain_probe(*t, q, rho)
            return (a_ <= margin, c_ if b_ <= margin else 0)
        if any(outcome((x + dx, y + dy, z + dz)) != outcome((x, y, z)) for dx in (-2e-5, 2e-5) for dy in (-2e-5, 2e-5) for dz in (-2e-5, 2e-5)):
            continue                                             # a tie of the rule itself (two ways out equally long, ...)
        byn = lambda c: (tuple(float(v) for v in c[1]), c[0])
        got = sorted(zip(o.arr("con_dist").tolist(), [tuple(r[:3]) for r in o.arr("con_frame").reshape(-1, 9).tolist()]), key=byn)
        got = got[::2] if len(got) % 2 == 0 else got            # the degenerate capsule reports every contact twice
        assert len(got) == len(want), f"probe {(x, y, z)}: oracle {got}, rule {want}"
        for (dg, ng), (dw_, nw) in zip(got, sorted(want, key=byn)):
            assert dg == pytest.approx(dw_, abs=tol) and tuple(ng) == tuple(float(v) for v in nw), f"probe {(x, y, z)}"
        n_wall += sum(1 for _, nrm in want if nrm[2] == 0)
        n_top += sum(1 for _, nrm in want if nrm[2] == 1)
    assert n_wall >= 25 and n_top >= 40, (n_wall, n_top)


@pytest.mark.parametrize("precision,rtol", [("f64", 1e-6), ("f32", 3e-2)])
def test_sphere_pushed_sideways_into_a_block_stops_at_its_face(oracle_lib, precision, rtol):
    """Known answer for the side faces: gravity with a horizontal component pushes a sphere that rests on a low cell
    against the raised cell next to it.  It stops at the face: the face's two contacts carry the push, the floor's two the
    weight, each at the penetration the documented soft-contact formulas give (tests/test_oracle_closed_form.py) — and
    the sphere stays on its side of the face.  With heightfield semantics (round 2) the same push carries it into the block."""
    import test_oracle_closed_form as cf
    from flygym_amd.compiler.model import EngineSemantics
    from tiny_models import sphere_on_plane

    gx = 0.45 * cf.G                                   # below the friction threshold: the sphere creeps, then leans on the face
    kw = dict(radius=cf.RADIUS, mu=cf.MU, solref=cf.SOLREF, solimp=cf.SOLIMP, margin=cf.MARGIN, gravity=(-gx, 0.0, -cf.G),
              terrain=TERRAINS["blocks"], start_xy=(cf.RADIUS + cf.MARGIN + 2e-4, 0.65), start_height=cf.RADIUS + cf.MARGIN)
    o = oracle_lib.Oracle(sphere_on_plane(cf.MASS, **kw).to_blob(), precision)       # cell (0, 0) is low, (-1, 0) raised: face at x = 0
    o.step(3000)
    assert o.ints()["ncon"] == 4 and np.abs(o.qvel[:3]).max() < (1e-6 if precision == "f64" else 1e-2)
    nrm = o.arr("con_frame").reshape(-1, 9)[:, :3]
    dist = o.arr("con_dist")
    wall, top = nrm[:, 0] == 1.0, nrm[:, 2] == 1.0
    assert wall.sum() == 2 and top.sum() == 2
    np.testing.assert_allclose(dist[wall] - cf.MARGIN, cf.rest_position(cf.MASS * gx), rtol=rtol)
    np.testing.assert_allclose(dist[top] - cf.MARGIN, cf.rest_position(cf.MASS * cf.G), rtol=rtol)
    f = o.arr("efc_force").reshape(4, 4).sum(1)
    assert f[wall].sum() == pytest.approx(cf.MASS * gx, rel=max(rtol, 1e-7)) and f[top].sum() == pytest.approx(cf.MASS * cf.G, rel=max(rtol, 1e-7))
    assert o.qpos[0] == pytest.approx(cf.RADIUS + cf.MARGIN + cf.rest_position(cf.MASS * gx), abs=1e-6)
    # round 2's height field: nothing holds it — it keeps creeping towards the raised cell (whose top would throw it up
    # once its centre is under it)
    o2 = oracle_lib.Oracle(sphere_on_plane(cf.MASS, semantics=EngineSemantics(terrain_walls="heightfield"), **kw).to_blob(), precision)
    o2.step(3000)
    assert o2.qpos[0] < o.qpos[0] - 2e-3                # it keeps creeping past the plane of the face (0.009 mm/s)
