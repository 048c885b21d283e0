"""Mesh → rigid-body constants (volume, centre of mass, inertia, convex hull, capsule fit).

The reference hands STL files to MuJoCo's model compiler
(``src/flygym/compose/fly.py:507-543`` mesh assets with scale (1000, ±1000, 1000);
``:603-611`` one mesh geom with explicit ``mass`` per body; ``:585-589`` geoms
re-typed to ``capsule``).  MuJoCo is not available, so this module restates what
that compiler derives from a mesh, from its documented behaviour:

* volume / COM / inertia by summing signed tetrahedra over the triangle soup
  (mesh ``inertia="exact"`` semantics);
* principal axes of the unit-density inertia; the geom frame is the principal
  frame centred at the COM;
* primitive fitting with ``fitaabb=false``: the *equivalent inertia box* of the
  mesh (the box with the same mass and principal moments) sets the capsule size:
  ``radius = (bx + by) / 2``, ``half_length = max(0, bz - radius / 2)`` with the
  capsule along the principal axis of smallest moment;
* the convex hull (qhull, as MuJoCo) is what collides.
"""

from __future__ import annotations

from dataclasses import dataclass
from pathlib import Path

import numpy as np

__all__ = ["MeshData", "load_binary_stl", "derive_mesh_data", "capsule_from_inertia_box", "capsule_from_aabb",
           "capsule_inertia", "convex_mesh_data", "mirror_y"]


def load_binary_stl(path: str | Path) -> np.ndarray:
    """Return the triangle soup ``(n_tri, 3, 3)`` float64 of a binary STL."""
    raw = Path(path).read_bytes()
    n = int(np.frombuffer(raw[80:84], dtype="<u4")[0])
    rec = np.dtype([("normal", "<f4", 3), ("v", "<f4", (3, 3)), ("attr", "<u2")])
    if len(raw) < 84 + n * rec.itemsize:
        raise ValueError(f"{path}: truncated binary STL")
    tris = np.frombuffer(raw, dtype=rec, count=n, offset=84)["v"]
    return tris.astype(np.float64)


@dataclass
class MeshData:
    """Geometry constants of one (scaled) mesh, all in the mesh file's frame."""

    volume: float            # signed-tetrahedra volume (positive after orientation fix)
    com: np.ndarray          # (3,)
    inertia: np.ndarray      # (3,3) unit-density inertia about the COM
    hull_vertices: np.ndarray  # (nh,3) convex-hull vertices
    hull_faces: np.ndarray   # (nf,3) int32 indices into hull_vertices, outward CCW
    n_vertices: int
    n_faces: int
    hull_volume: float

    def principal(self):
        """Principal moments (descending) and the rotation whose columns are the axes."""
        w, v = np.linalg.eigh(self.inertia)
        order = np.argsort(-w)  # descending: axis 2 = smallest moment = long axis
        w, v = w[order], v[:, order]
        # deterministic, right-handed: make the largest component of axes 0,1 positive
        for k in (0, 1):
            j = int(np.argmax(np.abs(v[:, k])))
            if v[j, k] < 0:
                v[:, k] = -v[:, k]
        v[:, 2] = np.cross(v[:, 0], v[:, 1])
        return w, v


def _signed_volume_props(tris: np.ndarray):
    a, b, c = tris[:, 0], tris[:, 1], tris[:, 2]
    det = np.einsum("ij,ij->i", a, np.cross(b, c))  # 6 * signed tet volume
    vol = det.sum() / 6.0
    com = ((a + b + c) * det[:, None]).sum(axis=0) / (24.0 * vol)
    # second moments  ∫ x xᵀ dV  over each tetrahedron (origin, a, b, c)
    s = a + b + c
    cov = (
        np.einsum("i,ij,ik->jk", det, a, a)
        + np.einsum("i,ij,ik->jk", det, b, b)
        + np.einsum("i,ij,ik->jk", det, c, c)
        + np.einsum("i,ij,ik->jk", det, s, s)
    ) / 120.0
    cov_c = cov - vol * np.outer(com, com)
    inertia = np.trace(cov_c) * np.eye(3) - cov_c
    return vol, com, inertia


def derive_mesh_data(tris: np.ndarray, scale=(1.0, 1.0, 1.0)) -> MeshData:
    from scipy.spatial import ConvexHull

    tris = tris * np.asarray(scale, dtype=np.float64)[None, None, :]
    vol, com, inertia = _signed_volume_props(tris)
    if vol < 0:  # inward-facing (e.g. mirrored) triangles
        tris = tris[:, ::-1, :]
        vol, com, inertia = _signed_volume_props(tris)
    pts = np.unique(tris.reshape(-1, 3), axis=0)
    hull = ConvexHull(pts)
    hv = pts[hull.vertices]
    remap = -np.ones(len(pts), dtype=np.int64)
    remap[hull.vertices] = np.arange(len(hull.vertices))
    faces = remap[hull.simplices]
    # orient hull faces outward
    centre = hv.mean(axis=0)
    fa, fb, fc = hv[faces[:, 0]], hv[faces[:, 1]], hv[faces[:, 2]]
    flip = np.einsum("ij,ij->i", np.cross(fb - fa, fc - fa), fa - centre) < 0
    faces[flip] = faces[flip][:, ::-1]
    return MeshData(
        volume=float(vol), com=com, inertia=inertia, hull_vertices=hv,
        hull_faces=faces.astype(np.int32), n_vertices=len(pts), n_faces=len(tris),
        hull_volume=float(hull.volume),
    )


def mirror_y(m: MeshData) -> MeshData:
    """The same mesh under y → −y (right-side segments reuse left meshes,
    reference ``fly.py:516-543``)."""
    s = np.array([1.0, -1.0, 1.0])
    S = np.diag(s)
    return MeshData(
        volume=m.volume, com=m.com * s, inertia=S @ m.inertia @ S,
        hull_vertices=m.hull_vertices * s, hull_faces=m.hull_faces[:, ::-1].copy(),
        n_vertices=m.n_vertices, n_faces=m.n_faces, hull_volume=m.hull_volume,
    )


def inertia_box_half_sizes(moments_desc: np.ndarray, volume: float) -> np.ndarray:
    """Half sizes of the solid box with mass ``volume`` and the given principal moments."""
    i0, i1, i2 = moments_desc
    m = volume
    return 0.5 * np.sqrt(
        np.maximum(0.0, 6.0 * np.array([i1 + i2 - i0, i0 + i2 - i1, i0 + i1 - i2]) / m)
    )


def capsule_from_inertia_box(m: MeshData) -> tuple[float, float]:
    """(radius, half_length) of the capsule MuJoCo fits to a mesh with ``fitaabb=false``."""
    w, _ = m.principal()
    b = inertia_box_half_sizes(w, m.volume)
    radius = 0.5 * (b[0] + b[1])
    half = max(0.0, b[2] - 0.5 * radius)
    return float(radius), float(half)


def capsule_from_aabb(m: MeshData) -> tuple[float, float, np.ndarray]:
    """(radius, half_length, centre) of the capsule fitted with ``fitaabb=true``: the same size rule applied to the
    half sizes of the axis-aligned bounding box of the mesh in its principal frame; the primitive sits at the box centre."""
    _, R = m.principal()
    loc = (m.hull_vertices - m.com) @ R
    lo, hi = loc.min(axis=0), loc.max(axis=0)
    b = 0.5 * (hi - lo)
    radius = 0.5 * (b[0] + b[1])
    half = max(0.0, b[2] - 0.5 * radius)
    return float(radius), float(half), m.com + R @ (0.5 * (lo + hi))


def convex_mesh_data(m: MeshData) -> MeshData:
    """The same mesh with volume / COM / inertia taken over its convex hull (mesh ``inertia="convex"``)."""
    vol, com, inertia = _signed_volume_props(m.hull_vertices[m.hull_faces])
    if vol < 0:
        vol, com, inertia = _signed_volume_props(m.hull_vertices[m.hull_faces[:, ::-1]])
    return MeshData(volume=float(vol), com=com, inertia=inertia, hull_vertices=m.hull_vertices, hull_faces=m.hull_faces,
                    n_vertices=m.n_vertices, n_faces=m.n_faces, hull_volume=m.hull_volume)


def capsule_inertia(radius: float, half: float, mass: float) -> np.ndarray:
    """Principal moments (x, y, z=axis) of a solid capsule scaled to ``mass``."""
    r, h = radius, 2.0 * half
    v_cyl = np.pi * r * r * h
    v_sph = 4.0 / 3.0 * np.pi * r ** 3
    vol = v_cyl + v_sph
    m_cyl, m_sph = mass * v_cyl / vol, mass * v_sph / vol
    izz = 0.5 * m_cyl * r * r + 0.4 * m_sph * r * r
    # two hemispheres, each m_sph/2, COM 3r/8 from its flat face placed at ±h/2
    ixx_cyl = m_cyl * (3 * r * r + h * h) / 12.0
    ixx_sph = m_sph * (0.4 * r * r + 0.25 * h * h + 0.375 * h * r)
    ixx = ixx_cyl + ixx_sph
    return np.array([ixx, ixx, izz])
