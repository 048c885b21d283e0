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


def render_eye_frames(cam_pos, cam_mat, height, width, fov_deg, checker_size, ground_z, sky_rgb, ground_rgb,
                      spheres=(), sphere_rgb=()):
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
    mat = np.zeros((height, width), dtype=np.int64)           # 0 sky, 1/2 ground, 3.. spheres, -1 black
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
    for s, sp in enumerate(spheres):
        oc = (cam - np.asarray(sp[:3], dtype=f)).astype(f)
        b = (d @ oc).astype(f)
        cc = f(oc @ oc - f(sp[3]) * f(sp[3]))
        disc = b * b - cc
        with np.errstate(invalid="ignore"):
            ts = (-b - np.sqrt(np.maximum(disc, 0))).astype(f)
        ok = (disc > 0) & (ts > 0) & (ts < tbest)
        mat = np.where(ok, 3 + s, mat)
        tbest = np.where(ok, ts, tbest)
    mat = np.where(theta > f(3.14159265), -1, mat)
    palette = np.zeros((4 + max(len(sphere_rgb), 0), 3), dtype=np.uint8)      # [0] black, [1 + m] material m
    palette[1] = sky_rgb
    palette[2], palette[3] = ground_rgb[0], ground_rgb[1]
    for s, c in enumerate(sphere_rgb):
        palette[4 + s] = c
    return palette[mat + 1]
