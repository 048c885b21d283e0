"""Compound-eye vision on the GPU: an eye renderer fused with the ommatidia resample.

The reference renders camera images with MuJoCo / MJWarp (``rendering.py``, ``warp/rendering.py``) and flygym 1.x
turned the two eye-camera images into ommatidia readings; this snapshot keeps only the constants
(``src/flygym/assets/model/legacy/flygym1_config.yaml:141-173``).  Build-defined here:

* each eye is a camera attached to its eye segment at the legacy offset; the legacy Euler triple is read as rotations
  about the fixed parent axes x, y, z in that order — the model's ``eulerseq: XYZ`` (``mujoco_globals.yaml:4``; upper
  case = extrinsic in MJCF) — which makes the left eye look 27 degrees forward of straight left, the right eye
  symmetrically; the camera looks along its -z, +y is up;
* the lens is an equidistant fisheye: a pixel's ray makes the angle ``rho * fov / 2`` with the optical axis, ``rho``
  = distance from the image centre in units of half the image height, ``fov`` = 157 degrees (``fovy_per_eye``); the
  legacy pinhole + distortion-coefficient pipeline is not reproduced;
* the scene is the world's ground plane with the reference's checker texture (4 mm squares of grey 0.3 / 0.4,
  ``compose/world.py:234-248``), a uniform sky and up to 8 opaque spheres (visual objects), unlit flat colours;
* a raw frame has one ray per pixel, colours rounded to uint8; the ommatidia readings are ``Retina``'s resample of
  that frame — computed in the same kernel, so the 2 x 691 KB of raw frames per fly never touch HBM unless
  ``render_frames`` is called.
"""

from __future__ import annotations

import ctypes

import numpy as np

from . import _native
from .sensors import EYE_CAMERAS, FOVY_PER_EYE_DEG, Retina

__all__ = ["EyeRenderer", "Scene"]


class _EyeParams(ctypes.Structure):
    _fields_ = [
        ("height", ctypes.c_int32), ("width", ctypes.c_int32), ("fov_deg", ctypes.c_float),
        ("eye_seg", ctypes.c_int32 * 2), ("rel_pos", (ctypes.c_float * 3) * 2), ("rel_quat", (ctypes.c_float * 4) * 2),
        ("checker_size", ctypes.c_float),
        ("sky_rgb", ctypes.c_uint8 * 4), ("ground_rgb", (ctypes.c_uint8 * 4) * 2), ("sphere_rgb", (ctypes.c_uint8 * 4) * 8),
        ("n_spheres", ctypes.c_int32), ("spheres_per_world", ctypes.c_int32),
    ]


def _euler_xyz_extrinsic_quat(e) -> np.ndarray:
    """Quaternion (w, x, y, z) of R = Rz(e2) Ry(e1) Rx(e0)."""
    def q(axis, a):
        out = np.zeros(4); out[0] = np.cos(a / 2); out[1 + axis] = np.sin(a / 2); return out

    def mul(a, b):
        w1, x1, y1, z1 = a; w2, x2, y2, z2 = b
        return np.array([w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2, w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2,
                         w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2, w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2])

    return mul(q(2, e[2]), mul(q(1, e[1]), q(0, e[0])))


def _u8(rgb) -> tuple:
    return tuple(int(np.floor(float(c) * 255.0 + 0.5)) for c in rgb)


class Scene:
    """What the eyes see: ground checker, sky colour, spheres ``(x, y, z, radius)`` with colours (floats in [0, 1])."""

    def __init__(self, checker_size: float = 4.0, ground_rgb=((0.3, 0.3, 0.3), (0.4, 0.4, 0.4)), sky_rgb=(0.55, 0.7, 0.9),
                 spheres=(), sphere_rgb=()):
        self.checker_size = float(checker_size)
        self.ground_rgb = tuple(_u8(c) for c in ground_rgb)
        self.sky_rgb = _u8(sky_rgb)
        self.spheres = np.asarray(spheres, dtype=np.float32).reshape(-1, 4)
        self.sphere_rgb = tuple(_u8(c) for c in sphere_rgb)
        if len(self.spheres) != len(self.sphere_rgb) or len(self.spheres) > 8:
            raise ValueError("need one colour per sphere and at most 8 spheres")


class EyeRenderer:
    """Renders both compound eyes of every fly in a ``HIPSimulation`` from the poses of the last step."""

    def __init__(self, sim, fly_name: str, scene: Scene | None = None, retina: Retina | None = None,
                 fov_deg: float = FOVY_PER_EYE_DEG):
        import torch

        self.sim, self.scene, self.retina = sim, scene or Scene(), retina or Retina()
        fly = sim.world.fly_lookup[fly_name]
        names = [s.name for s in fly.get_bodysegs_order()]
        p = _EyeParams()
        p.height, p.width, p.fov_deg = self.retina.height, self.retina.width, float(fov_deg)
        self.cameras = []
        for e, (seg, (pos, euler)) in enumerate(EYE_CAMERAS.items()):
            quat = _euler_xyz_extrinsic_quat(euler)
            p.eye_seg[e] = names.index(seg)
            for i in range(3):
                p.rel_pos[e][i] = pos[i]
            for i in range(4):
                p.rel_quat[e][i] = quat[i]
            self.cameras.append((seg, np.asarray(pos, dtype=np.float64), quat))
        p.checker_size = self.scene.checker_size
        for i in range(3):
            p.sky_rgb[i] = self.scene.sky_rgb[i]
            p.ground_rgb[0][i], p.ground_rgb[1][i] = self.scene.ground_rgb[0][i], self.scene.ground_rgb[1][i]
            for s, c in enumerate(self.scene.sphere_rgb):
                p.sphere_rgb[s][i] = c[i]
        p.n_spheres, p.spheres_per_world = len(self.scene.spheres), 0
        self._params = p
        self._spheres = torch.as_tensor(self.scene.spheres, device=sim.device) if len(self.scene.spheres) else None

    def set_spheres(self, spheres) -> None:
        """Move the spheres: ``(n_spheres, 4)`` shared by all worlds or ``(n_worlds, n_spheres, 4)`` per world
        (torch tensor on the GPU or numpy); the count must match the scene's."""
        t = self.sim._torch
        s = t.as_tensor(spheres, dtype=t.float32, device=self.sim.device).contiguous()
        n = len(self.scene.spheres)
        if tuple(s.shape) == (n, 4):
            self._params.spheres_per_world = 0
        elif tuple(s.shape) == (self.sim.n_worlds, n, 4):
            self._params.spheres_per_world = 1
        else:
            raise ValueError(f"expected spheres of shape ({n}, 4) or ({self.sim.n_worlds}, {n}, 4), got {tuple(s.shape)}")
        self._spheres = s

    def _call(self, frames, omm):
        t = self.sim._torch
        id_map, pale, inv_norm, plan = self.retina._device_constants(t, self.sim.device)
        if plan is None:
            raise ValueError("the eye renderer needs a retina whose pixel count is a multiple of 16")
        _native.check(_native.lib().nmf_eye_render(
            self.sim._batch_h, ctypes.byref(self._params),
            self._spheres.data_ptr() if self._spheres is not None else None,
            id_map.data_ptr(), plan.data_ptr(), pale.data_ptr(), inv_norm.data_ptr(), self.retina.num_ommatidia,
            frames.data_ptr() if frames is not None else None, omm.data_ptr() if omm is not None else None, self.sim._stream()))

    def render(self):
        """Ommatidia readings ``(n_worlds, 2, num_ommatidia, 2)`` float32 (eye 0 = left)."""
        t = self.sim._torch
        omm = t.empty((self.sim.n_worlds, 2, self.retina.num_ommatidia, 2), dtype=t.float32, device=self.sim.device)
        self._call(None, omm)
        return omm

    def render_frames(self, with_readings: bool = False):
        """Raw eye frames ``(n_worlds, 2, height, width, 3)`` uint8 (and the readings if asked)."""
        t = self.sim._torch
        frames = t.empty((self.sim.n_worlds, 2, self.retina.height, self.retina.width, 3), dtype=t.uint8, device=self.sim.device)
        omm = t.empty((self.sim.n_worlds, 2, self.retina.num_ommatidia, 2), dtype=t.float32, device=self.sim.device) if with_readings else None
        self._call(frames, omm)
        return (frames, omm) if with_readings else frames
