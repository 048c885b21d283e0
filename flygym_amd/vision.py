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
* the scene is the world's ground with the reference's checker texture (4 mm squares of grey 0.3 / 0.4,
  ``compose/world.py:234-248``) — the flat plane, or on the gapped / blocks / mixed worlds the very height map the
  physics collides with, side walls included (``Scene.terrain_relief``) — a uniform sky, up to 8 opaque spheres (visual
  objects) and the fly's own body: every segment flygym 1.x did not hide from the eye cameras
  (``legacy/flygym1_config.yaml:147-161`` hides the head, the antennae, the front coxae, the thorax) as the capsule
  fitted to its mesh (``body_capsules``; the reference's batch renderer sees the whole model,
  ``warp/rendering.py:385-441``); unlit flat colours;
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
        ("wall_rgb", ctypes.c_uint8 * 4), ("body_rgb", ctypes.c_uint8 * 4),
        ("n_capsules", ctypes.c_int32), ("terrain_relief", ctypes.c_int32), ("rays_per_ommatidium", ctypes.c_int32),
    ]


# segments flygym 1.x hid from the eye cameras (legacy/flygym1_config.yaml:147-161, in this build's segment names)
HIDDEN_SEGMENTS = ("lf_coxa", "l_eye", "l_arista", "l_funiculus", "l_pedicel", "rf_coxa", "r_eye", "r_arista", "r_funiculus",
                   "r_pedicel", "c_head", "c_rostrum", "c_haustellum", "c_thorax")


def body_capsules(fly, hidden=HIDDEN_SEGMENTS):
    """The fly's own body as the eyes see it: for every segment not in ``hidden`` the capsule fitted to its mesh (the
    equivalent-inertia-box fit of the model compiler).  Returns ``(segment indices (K,), geometry (K, 7))``: end points
    p0, p1 in the segment frame and the radius."""
    from .compiler.mesh import capsule_from_inertia_box
    from .compiler.model import load_asset_pack, mesh_for_segment

    pack = load_asset_pack(fly.asset_pack_path)
    names = [s.name for s in fly.get_bodysegs_order()]
    seg, geom = [], []
    for i, n in enumerate(names):
        if n in hidden:
            continue
        md = mesh_for_segment(pack, n, fly.mesh_type.value, fly.mirror_left2right)
        r, half = capsule_from_inertia_box(md)
        axis = md.principal()[1][:, 2]
        seg.append(i)
        geom.append([*(md.com - half * axis), *(md.com + half * axis), r])
    return np.asarray(seg, dtype=np.int32), np.asarray(geom, dtype=np.float32).reshape(-1, 7)


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
                 spheres=(), sphere_rgb=(), wall_rgb=(0.2, 0.2, 0.2), body_rgb=(0.47, 0.35, 0.24), terrain_relief=True,
                 own_body=True):
        """``terrain_relief``: render the world's height map (gapped / blocks / mixed worlds) instead of a flat plane;
        ``own_body``: render the fly's visible segments (``body_capsules``)."""
        self.checker_size = float(checker_size)
        self.ground_rgb = tuple(_u8(c) for c in ground_rgb)
        self.sky_rgb = _u8(sky_rgb)
        self.wall_rgb, self.body_rgb = _u8(wall_rgb), _u8(body_rgb)
        self.terrain_relief, self.own_body = bool(terrain_relief), bool(own_body)
        self.spheres = np.asarray(spheres, dtype=np.float32).reshape(-1, 4)
        self.sphere_rgb = tuple(_u8(c) for c in sphere_rgb)
        if len(self.spheres) != len(self.sphere_rgb) or len(self.spheres) > 8:
            raise ValueError("need one colour per sphere and at most 8 spheres")


class EyeRenderer:
    """Renders both compound eyes of every fly in a ``HIPSimulation`` from the poses of the last step."""

    def __init__(self, sim, fly_name: str, scene: Scene | None = None, retina: Retina | None = None,
                 fov_deg: float = FOVY_PER_EYE_DEG, rays_per_ommatidium: int = 0):
        """``rays_per_ommatidium``: 0 (default) casts every pixel of the raw frame that lies in an ommatidium's cell — the
        readings are those of resampling the rendered frame, bit for bit; 16 casts sixteen of each cell's pixels (one in
        fifteen: ``oracle/sensors_oracle.py::sampled_pixels`` is the specification) and reports their mean — an
        approximation of the cell mean at a fifteenth of the rays.  :meth:`render_frames` needs 0."""
        import torch

        if rays_per_ommatidium not in (0, 16):
            raise ValueError("rays_per_ommatidium must be 0 (every pixel) or 16")

        if hasattr(sim, "for_fly"):          # a world with several flies: this fly's batch
            sim = sim.for_fly(fly_name)
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
            p.wall_rgb[i], p.body_rgb[i] = self.scene.wall_rgb[i], self.scene.body_rgb[i]
            for s, c in enumerate(self.scene.sphere_rgb):
                p.sphere_rgb[s][i] = c[i]
        p.n_spheres, p.spheres_per_world = len(self.scene.spheres), 0
        p.terrain_relief = int(self.scene.terrain_relief)
        p.rays_per_ommatidium = int(rays_per_ommatidium)
        self.capsule_seg, self.capsule_geom = body_capsules(fly) if self.scene.own_body else (np.zeros(0, np.int32), np.zeros((0, 7), np.float32))
        p.n_capsules = len(self.capsule_seg)
        self._cap_seg = torch.as_tensor(self.capsule_seg, device=sim.device) if p.n_capsules else None
        self._cap_geom = torch.as_tensor(self.capsule_geom, device=sim.device) if p.n_capsules else None
        if ctypes.sizeof(_EyeParams) != _native.lib().nmf_eye_params_size():
            raise _native.NativeError("nmf_eye_params layout mismatch between vision.py and libnmf_hip.so")
        self._params = p
        self._spheres = torch.as_tensor(self.scene.spheres, device=sim.device) if len(self.scene.spheres) else None
        # the visit plan: an explicit handle of the library (nmf_eye_plan_create — synchronous, once), so that every render call is
        # one stream-ordered kernel launch and a vision tick can be captured in a hipGraph
        id_map, pale, inv_norm, plan = self.retina._device_constants(torch, sim.device)
        if plan is None:
            raise ValueError("the eye renderer needs a retina whose pixel count is a multiple of 16")
        torch.cuda.synchronize(sim.device)
        self._plan_h = _native.lib().nmf_eye_plan_create(id_map.data_ptr(), plan.data_ptr(), pale.data_ptr(), inv_norm.data_ptr(),
                                                         p.height, p.width, p.fov_deg, self.retina.num_ommatidia, sim.device_index)
        if not self._plan_h:
            raise _native.NativeError(_native.lib().nmf_last_error().decode())

    def __del__(self):
        try:
            if getattr(self, "_plan_h", None):
                _native.lib().nmf_eye_plan_destroy(self._plan_h)
                self._plan_h = None
        except Exception:
            pass

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
        _native.check(_native.lib().nmf_eye_render_planned(
            self.sim._batch_h, ctypes.byref(self._params), self._plan_h,
            self._spheres.data_ptr() if self._spheres is not None else None,
            self._cap_seg.data_ptr() if self._cap_seg is not None else None,
            self._cap_geom.data_ptr() if self._cap_geom is not None else None,
            frames.data_ptr() if frames is not None else None, omm.data_ptr() if omm is not None else None, self.sim._stream()))

    def render_into(self, omm):
        """:meth:`render` into a caller-owned ``(n_worlds, 2, num_ommatidia, 2)`` float32 tensor: no allocation, one kernel
        launch on the current stream — what a captured vision tick (``step(n)`` + this) replays."""
        t = self.sim._torch
        want = (self.sim.n_worlds, 2, self.retina.num_ommatidia, 2)
        if tuple(omm.shape) != want or omm.dtype != t.float32 or omm.device != self.sim.device or not omm.is_contiguous():
            raise ValueError(f"render_into needs a contiguous float32 {want} tensor on {self.sim.device}")
        self._call(None, omm)
        return omm

    def render(self):
        """Ommatidia readings ``(n_worlds, 2, num_ommatidia, 2)`` float32 (eye 0 = left)."""
        t = self.sim._torch
        omm = t.empty((self.sim.n_worlds, 2, self.retina.num_ommatidia, 2), dtype=t.float32, device=self.sim.device)
        self._call(None, omm)
        return omm

    def render_frames(self, with_readings: bool = False):
        """Raw eye frames ``(n_worlds, 2, height, width, 3)`` uint8 (and the readings if asked)."""
        t = self.sim._torch
        if self._params.rays_per_ommatidium:
            raise ValueError("the sampled mode renders no frames: build the EyeRenderer with rays_per_ommatidium=0")
        frames = t.empty((self.sim.n_worlds, 2, self.retina.height, self.retina.width, 3), dtype=t.uint8, device=self.sim.device)
        omm = t.empty((self.sim.n_worlds, 2, self.retina.num_ommatidia, 2), dtype=t.float32, device=self.sim.device) if with_readings else None
        self._call(frames, omm)
        return (frames, omm) if with_readings else frames
