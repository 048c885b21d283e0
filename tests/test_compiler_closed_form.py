"""Closed-form known answers for the model compiler's rigid-body constants (MuJoCo-free).

What MuJoCo's model compiler derives from a mesh is restated in ``flygym_amd/compiler/mesh.py`` from its documented
behaviour (exact mesh inertia by signed tetrahedra, principal geom frame, primitive fitting through the equivalent inertia
box, convex hull for collision).  Solids with textbook volume / centre of mass / inertia pin it here: a box, a box moved and
rotated (parallel axes), an inside-out triangle soup (mirrored meshes), a finely tessellated cylinder and sphere, and the
capsule whose own inertia formula closes the loop with the fit.
"""

import numpy as np
import pytest

from flygym_amd.compiler import mesh as M


def box_tris(a, b, c, centre=(0, 0, 0), R=np.eye(3)):
    """Triangle soup (12 outward triangles) of the box [-a, a] x [-b, b] x [-c, c], rotated by R and moved to ``centre``."""
    v = np.array([[sx * a, sy * b, sz * c] for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)], dtype=np.float64)
    quads = [(0, 1, 3, 2), (4, 6, 7, 5), (0, 4, 5, 1), (2, 3, 7, 6), (0, 2, 6, 4), (1, 5, 7, 3)]      # outward, counter-clockwise
    tris = []
    for q in quads:
        tris += [[v[q[0]], v[q[1]], v[q[2]]], [v[q[0]], v[q[2]], v[q[3]]]]
    t = np.array(tris)
    return t @ np.asarray(R).T + np.asarray(centre, dtype=np.float64)


def rot(axis, angle):
    axis = np.asarray(axis, dtype=np.float64) / np.linalg.norm(axis)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + np.sin(angle) * K + (1 - np.cos(angle)) * K @ K


def test_box_volume_com_inertia_and_hull():
    a, b, c = 0.3, 0.2, 0.7
    md = M.derive_mesh_data(box_tris(a, b, c))
    assert md.volume == pytest.approx(8 * a * b * c, rel=1e-12) and np.abs(md.com).max() < 1e-15
    want = md.volume / 3.0 * np.diag([b * b + c * c, a * a + c * c, a * a + b * b])      # unit density: m / 3 (b^2 + c^2) ...
    np.testing.assert_allclose(md.inertia, want, atol=1e-15)
    assert md.hull_vertices.shape == (8, 3) and md.hull_volume == pytest.approx(md.volume, rel=1e-12)
    w, V = md.principal()
    np.testing.assert_allclose(w, np.sort(np.diag(want))[::-1], rtol=1e-12)
    assert abs(abs(V[2, 2]) - 1.0) < 1e-12                # long axis (smallest moment) = z
    # the equivalent inertia box of a box is the box
    np.testing.assert_allclose(np.sort(M.inertia_box_half_sizes(w, md.volume)), np.sort([a, b, c]), rtol=1e-12)


def test_moved_rotated_and_inside_out_meshes():
    a, b, c = 0.3, 0.2, 0.7
    R, p = rot((1, 2, 3), 0.7), np.array([1.5, -0.4, 2.0])
    md = M.derive_mesh_data(box_tris(a, b, c, centre=p, R=R))
    want = 8 * a * b * c / 3.0 * np.diag([b * b + c * c, a * a + c * c, a * a + b * b])
    assert md.volume == pytest.approx(8 * a * b * c, rel=1e-12)
    np.testing.assert_allclose(md.com, p, atol=1e-13)
    np.testing.assert_allclose(md.inertia, R @ want @ R.T, atol=1e-13)             # about the centre of mass, in the file's frame
    # inside-out triangles (what mirroring a mesh produces): same solid
    flipped = M.derive_mesh_data(box_tris(a, b, c, centre=p, R=R)[:, ::-1, :])
    assert flipped.volume == pytest.approx(md.volume, rel=1e-12)
    np.testing.assert_allclose(flipped.inertia, md.inertia, atol=1e-13)
    # scale (the reference loads its meshes with scale 1000, and -1000 in y for the right side: compose/fly.py:507-543)
    s = np.array([1000.0, -1000.0, 1000.0])
    scaled = M.derive_mesh_data(box_tris(a, b, c), scale=s)
    assert scaled.volume == pytest.approx(1e9 * 8 * a * b * c, rel=1e-12)
    np.testing.assert_allclose(scaled.inertia, 1e15 * np.diag(np.diag(want)), rtol=1e-12)
    # the mirrored copy of a mesh is what mirror_y says it is
    tilted = M.derive_mesh_data(box_tris(a, b, c, centre=p, R=R))
    mirrored = M.derive_mesh_data(box_tris(a, b, c, centre=p, R=R) * np.array([1.0, -1.0, 1.0]))
    my = M.mirror_y(tilted)
    np.testing.assert_allclose(my.com, mirrored.com, atol=1e-13)
    np.testing.assert_allclose(my.inertia, mirrored.inertia, atol=1e-13)


def lathe(profile, n=720):
    """Closed surface of revolution about z from a (r, z) polyline whose ends sit on the axis."""
    ang = np.linspace(0, 2 * np.pi, n, endpoint=False)
    ring = lambda r, z, k: np.array([r * np.cos(ang[k % n]), r * np.sin(ang[k % n]), z])
    tris = []
    for (r0, z0), (r1, z1) in zip(profile[:-1], profile[1:]):
        for k in range(n):
            p00, p01, p10, p11 = ring(r0, z0, k), ring(r0, z0, k + 1), ring(r1, z1, k), ring(r1, z1, k + 1)
            if r0 > 0:
                tris.append([p00, p01, p11])
            if r1 > 0:
                tris.append([p00, p11, p10])
    return np.array(tris)


def test_cylinder_sphere_and_capsule_fit():
    r, h = 0.25, 0.9                                        # radius, half length
    cyl = M.derive_mesh_data(lathe([(0, -h), (r, -h), (r, h), (0, h)]))
    vol = 2 * h * np.pi * r * r
    assert abs(cyl.volume) == pytest.approx(vol, rel=1e-4)
    np.testing.assert_allclose(np.diag(cyl.inertia) / abs(cyl.volume), [r * r / 4 + h * h / 3, r * r / 4 + h * h / 3, r * r / 2], rtol=2e-4)
    # sphere: 2 / 5 m R^2
    th = np.linspace(-np.pi / 2, np.pi / 2, 181)
    sph = M.derive_mesh_data(lathe([(max(0.0, 0.4 * np.cos(t)) if 0 < i < 180 else 0.0, 0.4 * np.sin(t)) for i, t in enumerate(th)], n=360))
    assert abs(sph.volume) == pytest.approx(4 / 3 * np.pi * 0.4 ** 3, rel=5e-4)
    np.testing.assert_allclose(np.diag(sph.inertia) / abs(sph.volume), 0.4 * 0.4 * 0.4 * np.ones(3), rtol=1e-3)
    # capsule fit through the equivalent inertia box: a box b x b x L fits radius b and half length L - b / 2 (documented rule)
    b, L = 0.1, 0.8
    rad, half = M.capsule_from_inertia_box(M.derive_mesh_data(box_tris(b, b, L)))
    assert rad == pytest.approx(b, rel=1e-12) and half == pytest.approx(L - b / 2, rel=1e-12)
    # the capsule's own inertia: cylinder + two hemispheres, textbook formula
    m = 2.0
    I = M.capsule_inertia(r, h, m)
    vc, vs = 2 * h * np.pi * r * r, 4 / 3 * np.pi * r ** 3
    mc, ms = m * vc / (vc + vs), m * vs / (vc + vs)
    izz = mc * r * r / 2 + ms * 2 * r * r / 5
    ixx = mc * (r * r / 4 + h * h / 3) + ms * (2 * r * r / 5 + h * h + 3 * h * r / 4)
    np.testing.assert_allclose(np.diag(I) if np.ndim(I) == 2 else I[:3], [ixx, ixx, izz], rtol=1e-12)


def test_invweight0_and_mean_inertia_from_an_independent_route(bench_model, oracle_lib):
    """``seg_invweight0`` (what scales every contact's regulariser R) and ``stat_meaninertia`` (what scales the solver's
    tolerance) are MuJoCo's documented quantities: tr(J M^-1 J^T) / 3 at the segment's centre of mass for translation and
    rotation, and the mean diagonal of M, at qpos0.  The compiler computes them with its numpy kinematics
    (compiler/rigid.py); here they are rebuilt from the C oracle's mass matrix (CRBA) and Jacobians taken by finite
    differences of the oracle's forward kinematics — no code shared with the compiler."""
    fly, world, m = bench_model
    o = oracle_lib.Oracle(m.to_blob(), "f64")
    o.qpos[:] = m["qpos0"]
    o.forward()
    nv, nb = m.nv, m.nb
    M = o.arr("M").reshape(nv, nv).copy()
    assert float(m["stat_meaninertia"][0]) == pytest.approx(np.mean(np.diag(M)), rel=1e-9)
    Minv = np.linalg.inv(M)
    seg_body, body_ipos = np.asarray(m["seg_body"]), np.asarray(m["body_ipos"]).reshape(nb, 3)
    counts = np.bincount(seg_body, minlength=nb)

    def quat_to_mat(q):
        w, x, y, z = q
        return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                         [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                         [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])

    def pose(seg):
        p = o.arr("seg_xpos").reshape(-1, 3)[seg].copy()
        q = o.arr("seg_xquat").reshape(-1, 4)[seg].copy()
        return p, quat_to_mat(q / np.linalg.norm(q))

    checked = 0
    for seg in range(len(seg_body)):
        b = int(seg_body[seg])
        if b == 0 or counts[b] != 1 or np.abs(np.asarray(m["seg_pos"]).reshape(-1, 3)[seg]).max() > 0:
            continue                       # a segment that IS its dynamic body: its centre of mass is the body's
        if not fly_is_tarsus_or_tibia(m, seg):
            continue
        o.qpos[:] = m["qpos0"]; o.forward()
        p0, R0 = pose(seg)
        com0 = p0 + R0 @ body_ipos[b]
        root_p, root_R = o.qpos[:3].copy(), quat_to_mat(o.qpos[3:7] / np.linalg.norm(o.qpos[3:7]))
        J = np.zeros((6, nv))
        J[3:6, 0:3] = np.eye(3)                                   # root translation
        for i in range(3):                                        # root rotation about the root's own axes
            ax = root_R[:, i]
            J[0:3, 3 + i] = ax
            J[3:6, 3 + i] = np.cross(ax, com0 - root_p)
        eps = 1e-6
        for j in range(6, nv):
            o.qpos[:] = m["qpos0"]; o.qpos[j + 1] += eps; o.forward()
            p1, R1 = pose(seg)
            o.qpos[:] = m["qpos0"]; o.qpos[j + 1] -= eps; o.forward()
            p2, R2 = pose(seg)
            J[3:6, j] = ((p1 + R1 @ body_ipos[b]) - (p2 + R2 @ body_ipos[b])) / (2 * eps)
            dR = R1 @ R2.T                                          # rotation by 2 eps about the joint axis (if an ancestor)
            J[0:3, j] = np.array([dR[2, 1] - dR[1, 2], dR[0, 2] - dR[2, 0], dR[1, 0] - dR[0, 1]]) / (4 * eps)
        A = J @ Minv @ J.T
        want = np.asarray(m["seg_invweight0"]).reshape(-1, 2)[seg]
        assert np.trace(A[3:6, 3:6]) / 3 == pytest.approx(want[0], rel=1e-5), f"segment {seg}: translational"
        assert np.trace(A[0:3, 0:3]) / 3 == pytest.approx(want[1], rel=1e-5), f"segment {seg}: rotational"
        checked += 1
    assert checked >= 12


def fly_is_tarsus_or_tibia(m, seg):
    """Keep the test short: the distal leg segments (tibia and the five tarsal segments of each leg: the ones that touch the
    ground) — bodies 7 levels and more down their chains."""
    parent = np.asarray(m["body_parent"]); b = int(np.asarray(m["seg_body"])[seg]); depth = 0
    while b > 0:
        b = int(parent[b]); depth += 1
    return depth >= 6
