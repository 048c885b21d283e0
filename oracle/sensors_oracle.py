"""numpy restatement of the vision / olfaction sensor kernels.  TEST INFRASTRUCTURE.

The reference holds no implementation of these sensors (constants only:
src/flygym/assets/model/legacy/flygym1_config.yaml:141-192), so parity is to this build-defined
specification (see flygym_amd/sensors.py); PARITY UNPINNED against flygym 1.x.
"""

import numpy as np


def retina_resample(images: np.ndarray, id_map: np.ndarray, pale_mask: np.ndarray, inv_norm: np.ndarray) -> np.ndarray:
    """images (..., H, W, 3) uint8 -> (..., n_omm, 2) float32; integer sums, one float32 multiply."""
    n_omm = int(pale_mask.shape[0])
    lead = images.shape[:-3]
    flat = images.reshape((-1, id_map.size, 3))
    ids = id_map.ravel().astype(np.int64)
    chan = np.where(ids > 0, np.where(pale_mask[np.maximum(ids, 1) - 1] != 0, 2, 1), 1)
    out = np.zeros((flat.shape[0], n_omm, 2), dtype=np.float32)
    for k in range(flat.shape[0]):
        vals = flat[k][np.arange(ids.size), chan].astype(np.int64)
        sums = np.bincount(ids, weights=None, minlength=n_omm + 1) * 0
        sums = np.bincount(ids, weights=vals, minlength=n_omm + 1)[1:].astype(np.uint32)
        reading = sums.astype(np.float32) * inv_norm.astype(np.float32)
        out[k, :, 0] = np.where(pale_mask != 0, 0.0, reading)
        out[k, :, 1] = np.where(pale_mask != 0, reading, 0.0)
    return out.reshape(lead + (n_omm, 2))


def sampled_pixels(id_map: np.ndarray, n_ommatidia: int, rays: int = 16) -> np.ndarray:
    """The eye renderer's sampled mode (``nmf_eye_params.rays_per_ommatidium = 16``): for every ommatidium the pixels of
    its cell in raster order, P_0 .. P_{n-1}, and of those the ones at indices floor((2 j + 1) n / (2 rays)), j = 0 .. rays - 1
    (evenly spread over the cell, repeated where a cell has fewer pixels than rays).  Returns ``(n_ommatidia, rays)`` flat
    pixel indices, -1 for an ommatidium without pixels."""
    ids = id_map.ravel().astype(np.int64) & 0x7FFF
    out = np.full((n_ommatidia, rays), -1, dtype=np.int64)
    order = np.argsort(ids, kind="stable")
    counts = np.bincount(ids, minlength=n_ommatidia + 1)
    starts = np.concatenate([[0], np.cumsum(counts)])
    j = np.arange(rays)
    for o in range(n_ommatidia):
        n = int(counts[o + 1])
        if n:
            out[o] = order[starts[o + 1] + ((2 * j + 1) * n) // (2 * rays)]
    return out


def retina_sampled(images: np.ndarray, id_map: np.ndarray, pale_mask: np.ndarray, rays: int = 16) -> np.ndarray:
    """images (..., H, W, 3) uint8 -> (..., n_omm, 2) float32: the mean of the sampled pixels' colour byte (green for
    yellow-type ommatidia, blue for pale ones) / 255 — integer sum, one float32 multiply by 1 / (255 rays)."""
    n_omm = int(pale_mask.shape[0])
    px = sampled_pixels(id_map, n_omm, rays)
    lead = images.shape[:-3]
    flat = images.reshape((-1, id_map.size, 3))
    chan = np.where(pale_mask != 0, 2, 1)
    out = np.zeros((flat.shape[0], n_omm, 2), dtype=np.float32)
    scale = np.float32(1.0) / (np.float32(255.0) * np.float32(rays))
    for k in range(flat.shape[0]):
        vals = flat[k][np.maximum(px, 0), chan[:, None]].astype(np.int64)
        sums = np.where(px >= 0, vals, 0).sum(axis=1).astype(np.uint32)
        reading = sums.astype(np.float32) * scale
        out[k, :, 0] = np.where(pale_mask != 0, 0.0, reading)
        out[k, :, 1] = np.where(pale_mask != 0, reading, 0.0)
    return out.reshape(lead + (n_omm, 2))


def quat_to_mat(q):
    w, x, y, z = q
    return np.array([
        [1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
        [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
        [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def odor_intensity(seg_xpos, seg_xquat, sensor_seg, sensor_rel, src_pos, src_peak):
    """(n_worlds, nseg, 3/4) poses -> (n_worlds, n_dims, n_sensors) float64."""
    nw = seg_xpos.shape[0]
    out = np.zeros((nw, src_peak.shape[1], len(sensor_seg)))
    for w in range(nw):
        for k, sg in enumerate(sensor_seg):
            p = seg_xpos[w, sg] + quat_to_mat(seg_xquat[w, sg]) @ sensor_rel[k]
            d2 = ((p[None, :] - src_pos) ** 2).sum(axis=1)
            out[w, :, k] = (src_peak / d2[:, None]).sum(axis=0)
    return out


def euler_xyz_extrinsic_to_mat(e):
    """R = Rz(e[2]) Ry(e[1]) Rx(e[0]) — rotations about the fixed parent axes x, then y, then z."""
    cx, sx, cy, sy, cz, sz = np.cos(e[0]), np.sin(e[0]), np.cos(e[1]), np.sin(e[1]), np.cos(e[2]), np.sin(e[2])
    rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return rz @ ry @ rx


MAX_TERRAIN_CELLS = 64       # cells a ray is followed through the relief before the far field is taken as flat
TERRAIN_EPS = 1e-4           # the cell a ray is in at parameter t is the one that holds its point at t + eps (mm)
TERRAIN_WALL_TOL = 1e-4      # a cell is entered through its side wall if the ray is this far below its level (levels differ by >= 0.3 mm)


def terrain_cell(kind, p, x, y):
    """(x0, x1, y0, y1, h) of the constant-height cell of the build-defined terrains (flygym_amd/compose/world.py,
    the same h(x, y) the physics collides with) that holds (x, y); float32, vectorised.  Unbounded sides are +-inf."""
    f = np.float32
    x, y = np.asarray(x, dtype=f), np.asarray(y, dtype=f)
    inf = np.full(x.shape, np.inf, dtype=f)

    def gapped(block, gap, depth):
        period = f(block + gap)
        k = np.floor(x / period).astype(f)
        base = (k * period).astype(f)
        u = (x - base).astype(f)
        on = u < f(block)
        return (np.where(on, base, base + f(block)).astype(f), np.where(on, base + f(block), base + period).astype(f),
                -inf, inf, np.where(on, f(0), f(-depth)).astype(f))

    def blocks(size, height):
        i, j = np.floor(x / f(size)).astype(f), np.floor(y / f(size)).astype(f)
        ssum = i + j
        par = ssum - f(2) * np.floor(ssum / f(2))
        return ((i * f(size)).astype(f), ((i + 1) * f(size)).astype(f), (j * f(size)).astype(f), ((j + 1) * f(size)).astype(f),
                np.where(par != 0, f(height), f(0)).astype(f))

    if kind == 1:
        return gapped(p[0], p[1], p[2])
    if kind == 2:
        return blocks(p[0], p[1])
    if kind == 3:
        st = np.floor(x / f(p[3])).astype(f)
        k = st - f(3) * np.floor(st / f(3))
        s0, s1 = (st * f(p[3])).astype(f), ((st + 1) * f(p[3])).astype(f)
        g, bk = gapped(1.0, p[1], p[2]), blocks(p[0], 0.35)
        x0 = np.where(k == 1, np.maximum(g[0], s0), np.where(k == 2, np.maximum(bk[0], s0), s0)).astype(f)
        x1 = np.where(k == 1, np.minimum(g[1], s1), np.where(k == 2, np.minimum(bk[1], s1), s1)).astype(f)
        y0 = np.where(k == 2, bk[2], -inf).astype(f)
        y1 = np.where(k == 2, bk[3], inf).astype(f)
        h = np.where(k == 1, g[4], np.where(k == 2, bk[4], f(0))).astype(f)
        return x0, x1, y0, y1, h
    return -inf, inf, -inf, inf, np.zeros(x.shape, dtype=f)


def _ray_capsule(d, pa, pb, r):
    """Nearest positive hit parameter of unit rays d (.., 3) from the origin with the capsule (pa, pb, r); inf = miss."""
    f = np.float32
    pa, pb = np.asarray(pa, dtype=f), np.asarray(pb, dtype=f)
    ba = (pb - pa).astype(f)
    baba, baoa, oaoa = f(ba @ ba), f(-(ba @ pa)), f(pa @ pa)
    bard = (d @ ba).astype(f)
    rdoa = (-(d @ pa)).astype(f)
    a = (baba - bard * bard).astype(f)
    b = (baba * rdoa - baoa * bard).astype(f)
    c = f(baba * oaoa - baoa * baoa - f(r) * f(r) * baba)
    h = (b * b - a * c).astype(f)
    with np.errstate(divide="ignore", invalid="ignore"):
        t = ((-b - np.sqrt(np.maximum(h, 0))) / a).astype(f)
    yy = (baoa + t * bard).astype(f)
    body = (h >= 0) & (a > f(1e-12)) & (yy > 0) & (yy < baba) & (t > 0)
    # end caps: the sphere at pa when the axial coordinate is <= 0 (or the ray misses the infinite cylinder), else at pb
    use_a = ~(yy > 0) | ~((h >= 0) & (a > f(1e-12)))
    ocx = np.where(use_a[..., None], -pa, -pb).astype(f)
    bb = (d * ocx).sum(axis=-1).astype(f)
    cc = ((ocx * ocx).sum(axis=-1) - f(r) * f(r)).astype(f)
    hh = (bb * bb - cc).astype(f)
    with np.errstate(invalid="ignore"):
        tc = (-bb - np.sqrt(np.maximum(hh, 0))).astype(f)
    cap = (hh > 0) & (tc > 0)
    return np.where(body, t, np.where(cap, tc, np.inf)).astype(f)


def render_eye_frames(cam_pos, cam_mat, height, width, fov_deg, checker_size, ground_z, sky_rgb, ground_rgb,
                      spheres=(), sphere_rgb=(), terrain=None, wall_rgb=(51, 51, 51), capsules=(), body_rgb=(120, 90, 60)):
    """Raw eye frame (height, width, 3) uint8 of one camera — the specification of csrc/nmf_eyes.hip.

    Camera frame: x right, y up, looking along -z; ``cam_mat`` columns are those axes in world coordinates.
    Equidistant fisheye: the ray through pixel centre (row i, col j) makes the angle ``rho * fov/2`` with the optical
    axis, ``rho`` = distance from the image centre in units of half the image height.  Scene: ground plane z =
    ground_z textured with checker squares of side ``checker_size`` (parity of floor(x/s) + floor(y/s)), uniform sky,
    opaque spheres (x, y, z, radius); the nearest hit wins; pixels beyond a polar angle of pi are black."""
    f = np.float32
    i, j = np.mgrid[0:height, 0:width]
    u = ((j.astype(f) + f(0.5) - f(0.5 * width)) * f(2.0 / height)).astype(f)
    v = ((i.astype(f) + f(0.5) - f(0.5 * height)) * f(2.0 / height)).astype(f)
    rho2 = u * u + v * v
    rinv = (1.0 / np.sqrt(np.maximum(rho2, f(1e-12)))).astype(f)
    theta = (rho2 * rinv * f(0.5 * fov_deg * np.pi / 180.0)).astype(f)
    st, ct = np.sin(theta).astype(f), np.cos(theta).astype(f)
    dcam = np.stack([st * u * rinv, -st * v * rinv, -ct], axis=-1).astype(f)
    R = np.asarray(cam_mat, dtype=f)
    d = (dcam @ R.T).astype(f)
    cam = np.asarray(cam_pos, dtype=f)
    mat = np.zeros((height, width), dtype=np.int64)           # 0 sky, 1/2 ground, 3 wall, 4 body, 5.. spheres, -1 black
    tbest = np.full((height, width), np.inf, dtype=f)
    hz = f(cam[2] - f(ground_z))
    with np.errstate(divide="ignore", invalid="ignore"):
        t = (-hz / d[..., 2]).astype(f)
    hit = (d[..., 2] < 0) & (t > 0)
    gx = ((cam[0] + t * d[..., 0]) * f(1.0 / checker_size)).astype(f)
    gy = ((cam[1] + t * d[..., 1]) * f(1.0 / checker_size)).astype(f)
    with np.errstate(invalid="ignore"):
        par = (np.floor(np.where(hit, gx, 0)).astype(np.int64) + np.floor(np.where(hit, gy, 0)).astype(np.int64)) & 1
    mat = np.where(hit, 1 + par, mat)
    tbest = np.where(hit, t, tbest)
    if terrain is not None and int(terrain[0]) != 0:
        # relief (terrain = (kind, params p0..p3, highest level)): follow every downward ray through the constant-height
        # cells of h(x, y) between the highest and the lowest level — side walls included — for at most MAX_TERRAIN_CELLS
        # cells; beyond that the ground keeps the flat-plane answer computed above
        kind, tp, hmax = int(terrain[0]), [float(v) for v in terrain[1]], f(terrain[2])
        zmin = f(ground_z) + f(min(0.0, -tp[2] if kind in (1, 3) else 0.0))
        down = d[..., 2] < 0
        with np.errstate(divide="ignore", invalid="ignore"):
            tcur = np.where(down, np.maximum(f(0), (f(ground_z) + hmax - cam[2]) / d[..., 2]), np.inf).astype(f)
        live = down.copy()
        t_hit = np.full((height, width), np.inf, dtype=f)
        m_hit = np.zeros((height, width), dtype=np.int64)
        for _ in range(MAX_TERRAIN_CELLS):
            tprobe = (tcur + f(TERRAIN_EPS)).astype(f)
            px = (cam[0] + tprobe * d[..., 0]).astype(f)
            py = (cam[1] + tprobe * d[..., 1]).astype(f)
            x0, x1, y0, y1, h = terrain_cell(kind, tp, np.where(live, px, 0), np.where(live, py, 0))
            h = (h + f(ground_z)).astype(f)
            z_in = (cam[2] + tcur * d[..., 2]).astype(f)
            wall = live & (z_in < h - f(TERRAIN_WALL_TOL))
            with np.errstate(divide="ignore", invalid="ignore"):
                tx = np.where(d[..., 0] > 0, (x1 - cam[0]) / d[..., 0], np.where(d[..., 0] < 0, (x0 - cam[0]) / d[..., 0], np.inf)).astype(f)
                ty = np.where(d[..., 1] > 0, (y1 - cam[1]) / d[..., 1], np.where(d[..., 1] < 0, (y0 - cam[1]) / d[..., 1], np.inf)).astype(f)
                t_h = ((h - cam[2]) / d[..., 2]).astype(f)
            t_out = np.minimum(tx, ty)
            top = live & ~wall & (t_h <= t_out)
            qx = ((cam[0] + t_h * d[..., 0]) * f(1.0 / checker_size)).astype(f)
            qy = ((cam[1] + t_h * d[..., 1]) * f(1.0 / checker_size)).astype(f)
            with np.errstate(invalid="ignore"):
                tpar = (np.floor(np.where(top, qx, 0)).astype(np.int64) + np.floor(np.where(top, qy, 0)).astype(np.int64)) & 1
            t_hit = np.where(wall, tcur, np.where(top, t_h, t_hit)).astype(f)
            m_hit = np.where(wall, 3, np.where(top, 1 + tpar, m_hit))
            live = live & ~wall & ~top
            tcur = np.where(live, t_out, tcur).astype(f)
        done = down & ~live
        mat = np.where(done, m_hit, mat)
        tbest = np.where(done, t_hit, tbest)
    for s, sp in enumerate(spheres):
        oc = (cam - np.asarray(sp[:3], dtype=f)).astype(f)
        b = (d @ oc).astype(f)
        cc = f(oc @ oc - f(sp[3]) * f(sp[3]))
        disc = b * b - cc
        with np.errstate(invalid="ignore"):
            ts = (-b - np.sqrt(np.maximum(disc, 0))).astype(f)
        ok = (disc > 0) & (ts > 0) & (ts < tbest)
        mat = np.where(ok, 5 + s, mat)
        tbest = np.where(ok, ts, tbest)
    for cap in capsules:                                       # the fly's own body: (p0, p1, radius) in world coordinates
        tcap = _ray_capsule(d, np.asarray(cap[0], dtype=f) - cam, np.asarray(cap[1], dtype=f) - cam, cap[2])
        ok = tcap < tbest
        mat = np.where(ok, 4, mat)
        tbest = np.where(ok, tcap, tbest)
    # materials: 0 sky, 1 / 2 ground checker, 3 terrain side wall, 4 own body, 5.. spheres, -1 black
    mat = np.where(theta > f(3.14159265), -1, mat)
    palette = np.zeros((6 + max(len(sphere_rgb), 0), 3), dtype=np.uint8)      # [0] black, [1 + m] material m
    palette[1] = sky_rgb
    palette[2], palette[3] = ground_rgb[0], ground_rgb[1]
    palette[4], palette[5] = wall_rgb, body_rgb
    for s, c in enumerate(sphere_rgb):
        palette[6 + s] = c
    return palette[mat + 1]
