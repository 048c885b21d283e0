"""Vision and olfaction sensors on the GPU.

The reference snapshot keeps only the constants of flygym 1.x's retina and odor sensors
(``src/flygym/assets/model/legacy/flygym1_config.yaml:141-192``): 721 ommatidia per eye, 512 x 450 raw
eye images, eye-camera placement, four odor sensor sites.  The ommatidia id map and pale-type mask files it
names (``:199-200``) are not shipped, and no code exists, so the following is *build-defined*:

* the id map is a flat-top hexagonal lattice of radius 15 (3*15*16 + 1 = 721 cells) whose circum-size makes
  it span the 512-pixel height (it then spans 448 of the 450 columns); a pixel belongs to the cell whose
  centre is nearest, or to none outside the hexagon; ids run row-major over cell centres;
* 30 % of the ommatidia are "pale" (read the blue channel), the rest "yellow" (read green), drawn once with
  ``numpy.random.default_rng(0)``;
* a reading is the mean of the cell's pixels in its channel / 255, stored in channel 0 (yellow) or 1 (pale);
* odor intensity = sum over sources of ``peak / distance**2`` at the four sensor sites.

The arithmetic runs in ``libnmf_hip.so`` (``nmf_retina_resample``, ``nmf_odor_intensity``).
"""

from __future__ import annotations

import ctypes

import numpy as np

from . import _native

__all__ = ["Retina", "OdorSensors", "RAW_IMG_HEIGHT", "RAW_IMG_WIDTH", "NUM_OMMATIDIA"]

RAW_IMG_HEIGHT = 512     # flygym1_config.yaml:143
RAW_IMG_WIDTH = 450      # :144
NUM_OMMATIDIA = 721      # :145
FOVY_PER_EYE_DEG = 157   # :142
EYE_CAMERAS = {          # :164-173  (parent segment, offset mm, euler orientation)
    "l_eye": ((-0.03, 0.38, 0.0), (1.57, 0.00, -0.47)),
    "r_eye": ((-0.03, -0.38, 0.0), (-1.57, 3.14, 0.47)),
}
ODOR_SENSOR_SITES = [    # :175-192  (parent segment, offset in the segment frame, mm)
    ("c_rostrum", (-0.15, 0.15, -0.15)),     # left maxillary palp
    ("c_rostrum", (-0.15, -0.15, -0.15)),    # right maxillary palp
    ("l_funiculus", (0.02, 0.00, -0.10)),    # left antenna
    ("r_funiculus", (0.02, 0.00, -0.10)),    # right antenna
]


def make_ommatidia_id_map(height: int = RAW_IMG_HEIGHT, width: int = RAW_IMG_WIDTH, radius: int = 15) -> np.ndarray:
    """(height, width) int16 map: 0 = no ommatidium, k = ommatidium k-1."""
    size = height / ((2 * radius + 1) * np.sqrt(3.0))          # hexagon spans the image height
    cells = [(q, r) for q in range(-radius, radius + 1) for r in range(-radius, radius + 1) if abs(q + r) <= radius]
    centres = np.array([(size * 1.5 * q, size * np.sqrt(3.0) * (r + 0.5 * q)) for q, r in cells])   # (x, y)
    order = np.lexsort((centres[:, 0], centres[:, 1]))          # row-major over centres: y, then x
    ident = {cells[i]: k + 1 for k, i in enumerate(order)}
    ys, xs = np.mgrid[0:height, 0:width]
    px = xs + 0.5 - 0.5 * width
    py = ys + 0.5 - 0.5 * height
    qf = (2.0 / 3.0) * px / size
    rf = (-1.0 / 3.0 * px + np.sqrt(3.0) / 3.0 * py) / size
    x, z = qf, rf
    y = -x - z
    rx, ry, rz = np.rint(x), np.rint(y), np.rint(z)
    dx, dy, dz = np.abs(rx - x), np.abs(ry - y), np.abs(rz - z)
    fix_x = (dx > dy) & (dx > dz)
    fix_y = ~fix_x & (dy > dz)
    rx = np.where(fix_x, -ry - rz, rx)
    rz = np.where(~fix_x & ~fix_y, -rx - ry, rz)
    q, r = rx.astype(int), rz.astype(int)
    inside = (np.abs(q) <= radius) & (np.abs(r) <= radius) & (np.abs(q + r) <= radius)
    out = np.zeros((height, width), dtype=np.int16)
    lut = np.zeros((2 * radius + 1, 2 * radius + 1), dtype=np.int16)
    for (cq, cr), k in ident.items():
        lut[cq + radius, cr + radius] = k
    out[inside] = lut[q[inside] + radius, r[inside] + radius]
    return out


class Retina:
    """Compound-eye model: raw eye images -> ommatidia readings, batched on the GPU."""

    def __init__(self, id_map: np.ndarray | None = None, pale_mask: np.ndarray | None = None, pale_fraction: float = 0.3):
        self.id_map = make_ommatidia_id_map() if id_map is None else np.ascontiguousarray(id_map, dtype=np.int16)
        self.height, self.width = self.id_map.shape
        self.num_ommatidia = int(self.id_map.max())
        counts = np.bincount(self.id_map.ravel(), minlength=self.num_ommatidia + 1)[1:]
        if (counts == 0).any():
            raise ValueError("every ommatidium needs at least one pixel")
        self.num_pixels_per_ommatidium = counts
        if pale_mask is None:
            pale_mask = np.random.default_rng(0).random(self.num_ommatidia) < pale_fraction
        self.pale_mask = np.ascontiguousarray(pale_mask, dtype=np.uint8)
        self.inv_norm = (1.0 / (255.0 * counts)).astype(np.float32)
        self._dev = None

    def _device_constants(self, torch, device):
        if self._dev is None or self._dev[0] != device:
            ids = self.id_map.ravel().astype(np.int64)
            typed = ids | (np.where(ids > 0, self.pale_mask[np.maximum(ids, 1) - 1], 0).astype(np.int64) << 15)
            id_dev = torch.as_tensor(typed.astype(np.uint16).view(np.int16), device=device)
            plan = None
            if typed.size % 16 == 0:      # run plan of the id map (streaming kernel); built once per device
                plan = torch.empty(_native.lib().nmf_retina_plan_bytes(typed.size), dtype=torch.uint8, device=device)
                stream = ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)
                _native.check(_native.lib().nmf_retina_plan(id_dev.data_ptr(), typed.size, plan.data_ptr(), stream))
            self._dev = (device, id_dev, torch.as_tensor(self.pale_mask, device=device),
                         torch.as_tensor(self.inv_norm, device=device), plan)
        return self._dev[1:]

    def raw_image_to_hex_pxls(self, images):
        """``images``: torch uint8 tensor ``(..., height, width, 3)`` on the GPU ->
        float32 ``(..., num_ommatidia, 2)`` (channel 0 = yellow-type reading, 1 = pale-type reading)."""
        import torch

        if images.dtype != torch.uint8 or images.shape[-3:] != (self.height, self.width, 3):
            raise ValueError(f"expected uint8 images of shape (..., {self.height}, {self.width}, 3), got {tuple(images.shape)}")
        if not images.is_cuda:
            raise _native.NativeError("the retina resample runs on the MI355X: pass a GPU tensor")
        images = images.contiguous()
        lead = tuple(images.shape[:-3])
        n = int(np.prod(lead)) if lead else 1
        id_map, pale, inv_norm, plan = self._device_constants(torch, images.device)
        out = torch.empty(lead + (self.num_ommatidia, 2), dtype=torch.float32, device=images.device)
        stream = ctypes.c_void_p(torch.cuda.current_stream(images.device).cuda_stream)
        _native.check(_native.lib().nmf_retina_resample(
            images.data_ptr(), id_map.data_ptr(), plan.data_ptr() if plan is not None else None, pale.data_ptr(),
            inv_norm.data_ptr(), n,
            self.height * self.width, self.num_ommatidia, out.data_ptr(), stream))
        return out

    def hex_pxls_to_human_readable(self, readings: np.ndarray) -> np.ndarray:
        """Paint ``(num_ommatidia, 2)`` readings back onto the pixel grid (for inspection)."""
        vals = np.concatenate([[0.0], np.asarray(readings).max(axis=-1)])
        return vals[self.id_map]


class OdorSensors:
    """The fly's four odor sensors (two maxillary palps, two antennae) in an odor field of point sources."""

    def __init__(self, sim, fly_name: str, source_positions, peak_intensities):
        import torch

        if hasattr(sim, "for_fly"):          # a world with several flies: this fly's batch
            sim = sim.for_fly(fly_name)
        self.sim = sim
        fly = sim.world.fly_lookup[fly_name]
        names = [s.name for s in fly.get_bodysegs_order()]
        self.source_positions = np.asarray(source_positions, dtype=np.float32).reshape(-1, 3)
        self.peak_intensities = np.asarray(peak_intensities, dtype=np.float32).reshape(len(self.source_positions), -1)
        self.n_dims = self.peak_intensities.shape[1]
        dev = sim.device
        self._seg = torch.as_tensor(np.array([names.index(n) for n, _ in ODOR_SENSOR_SITES], dtype=np.int32), device=dev)
        self._rel = torch.as_tensor(np.array([r for _, r in ODOR_SENSOR_SITES], dtype=np.float32), device=dev)
        self._pos = torch.as_tensor(self.source_positions, device=dev)
        self._peak = torch.as_tensor(self.peak_intensities, device=dev)

    def get_odor_intensities(self):
        """float32 ``(n_worlds, n_dims, 4)`` from the poses of the last step."""
        import torch

        out = torch.empty((self.sim.n_worlds, self.n_dims, 4), dtype=torch.float32, device=self.sim.device)
        _native.check(_native.lib().nmf_odor_intensity(
            self.sim._batch_h, self._seg.data_ptr(), self._rel.data_ptr(), 4, self._pos.data_ptr(),
            self._peak.data_ptr(), len(self.source_positions), self.n_dims, out.data_ptr(), self.sim._stream()))
        return out
