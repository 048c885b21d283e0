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
