"""Worlds (mirror of reference ``src/flygym/compose/world.py:21-366``).

A world fixes where the fly spawns, what it can collide with and which contact sensors
exist.  ``compile()`` returns the flat :class:`~flygym_amd.compiler.model.CompiledModel`
instead of the reference's ``(MjModel, MjData)`` pair.
"""

from __future__ import annotations

from abc import ABC, abstractmethod
from typing import Any

import numpy as np

from ..anatomy import LEG_LINKS, BodySegment, ContactBodiesPreset
from ..utils.math import Rotation3D
from .fly import Fly
from .physics import ContactParams

__all__ = ["BaseWorld", "FlatGroundWorld", "TetheredWorld", "GappedTerrainWorld", "BlocksTerrainWorld",
           "MixedTerrainWorld"]


class BaseWorld(ABC):
    def __init__(self, name: str) -> None:
        self.name = name
        self._fly_lookup: dict[str, Fly] = {}
        self.world_dof_neutral_states: dict[str, list[float]] = {}
        self.noslip_iterations = 5  # mujoco_globals.yaml:15; GPU path sets 0
        self.spawn_position = np.zeros(3)
        self.spawn_quat = np.array([1.0, 0.0, 0.0, 0.0])
        self.bodysegs_with_ground_contact: list[BodySegment] = []
        self.ground_contact_params = ContactParams()
        self.add_ground_contact_sensors = False
        self.legpos_to_groundcontactsensors_by_fly = None
        self.fixed_base = False
        self.terrain_type = 0                  # 0 flat, 1 gapped, 2 blocks, 3 mixed (see terrain_height)
        self.terrain_params = (0.0, 0.0, 0.0, 0.0)
        from ..compiler.model import EngineSemantics

        self.semantics = EngineSemantics()     # engine-stage semantics MuJoCo would decide (see the class docstring)
        self._compiled = None

    @property
    def fly_lookup(self) -> dict[str, Fly]:
        return self._fly_lookup

    @abstractmethod
    def _attach_fly(self, fly: Fly, spawn_position, spawn_rotation: Rotation3D, *args, **kwargs) -> str:
        """Record how the fly hangs in this world; returns the free-joint name."""

    # what one fly's attachment sets on the world (the single-fly attributes the compiler reads); a world with several flies keeps
    # one record per fly and hands the compiler one fly at a time (single_fly_view)
    _ATTACHMENT_FIELDS = ("spawn_position", "spawn_quat", "bodysegs_with_ground_contact", "ground_contact_params",
                          "add_ground_contact_sensors", "legpos_to_groundcontactsensors_by_fly")

    def add_fly(self, fly: Fly, spawn_position, spawn_rotation: Rotation3D, *args: Any, **kwargs: Any) -> None:
        """Attach a fly (reference ``compose/world.py:95-149``).  A world takes several flies, as the reference's does.  The
        reference gives fly geoms ``contype = conaffinity = 0`` and adds explicit fly-ground pairs only (``compose/fly.py:609-610``,
        ``compose/world.py:300-309``): flies of one world never touch each other, so each is its own dynamical system — the
        simulation classes step one batch per fly (``simulation.py::MultiFlyHIPSimulation``)."""
        if fly.name in self._fly_lookup:
            raise ValueError(f"Fly with name '{fly.name}' already exists in the world.")
        if spawn_rotation.format != "quat":
            raise ValueError(
                "Freejoint neutral rotation can only be specified in quaternion format "
                f"for now. Got {spawn_rotation}."
            )
        sensors_so_far = dict(self.legpos_to_groundcontactsensors_by_fly or {})
        self._fly_lookup[fly.name] = fly
        freejoint = self._attach_fly(fly, spawn_position, spawn_rotation, *args, **kwargs)
        self.spawn_position = np.asarray(spawn_position, dtype=np.float64)
        self.spawn_quat = spawn_rotation.as_quat()
        self.world_dof_neutral_states[freejoint] = [*self.spawn_position, *spawn_rotation.values]
        if not hasattr(self, "_attachments"):
            self._attachments = {}
        self._attachments[fly.name] = {k: getattr(self, k) for k in self._ATTACHMENT_FIELDS}
        if self.legpos_to_groundcontactsensors_by_fly is not None:       # the by-fly lookup of the whole world (reference name)
            sensors_so_far.update(self.legpos_to_groundcontactsensors_by_fly)
            self.legpos_to_groundcontactsensors_by_fly = sensors_so_far
        self._compiled = None

    def single_fly_view(self, fly_name: str) -> "BaseWorld":
        """This world with ``fly_name`` as its only fly (a shallow copy: same terrain, options and semantics object): what the
        compiler and one batch of a multi-fly simulation see."""
        import copy

        if fly_name not in self._fly_lookup:
            raise KeyError(f"no fly named '{fly_name}' in world '{self.name}'")
        view = copy.copy(self)
        view._fly_lookup = {fly_name: self._fly_lookup[fly_name]}
        for k, v in self._attachments[fly_name].items():
            setattr(view, k, v)
        view.world_dof_neutral_states = {k: v for k, v in self.world_dof_neutral_states.items() if k == f"{fly_name}/"}
        view._compiled = None
        return view

    def compile_model(self, fly_name: str | None = None):
        """The engine's compiled model of this world (cached until the world changes).  A world with several flies compiles one
        model per fly (``fly_name``): its flies are independent dynamical systems (see :meth:`add_fly`)."""
        from ..compiler.model import compile_world

        if len(self._fly_lookup) > 1:
            if fly_name is None:
                raise ValueError(f"world '{self.name}' holds {len(self._fly_lookup)} flies: compile_model(fly_name) compiles one of them "
                                 f"({', '.join(self._fly_lookup)})")
            return self.single_fly_view(fly_name).compile_model()
        if fly_name is not None and fly_name not in self._fly_lookup:
            raise KeyError(f"no fly named '{fly_name}' in world '{self.name}'")

        # the cache is only as good as what it was compiled from: world.semantics is a plain mutable object, so a flag set
        # after the first compile must not be silently ignored
        if self._compiled is not None and self._compiled.meta.get("semantics") != self.semantics.as_dict():
            self._compiled = None
        if self._compiled is None:
            self._compiled = compile_world(self)
        return self._compiled

    def compile(self):
        """``(model, data)`` as the reference's ``BaseWorld.compile`` (``compose/base.py:21-27``) — here the engine's
        :class:`~flygym_amd.compiler.model.CompiledModel` (sizes ``nq nv nu nbody njnt nsite ncam``, arrays by name) and
        the keyframe state, not MuJoCo objects."""
        from ..compiler.model import CompiledData

        model = self.compile_model()
        return model, CompiledData(model)


class _FreeSpaceWorld(BaseWorld):
    """No ground, no tether: what ``Fly.compile()`` compiles a standalone fly in."""

    def __init__(self) -> None:
        super().__init__("free_space")

    def _attach_fly(self, fly, spawn_position, spawn_rotation) -> str:
        return f"{fly.name}/"


def _sort_prox2dist(segs: list[BodySegment]) -> list[BodySegment]:
    return sorted(segs, key=lambda s: LEG_LINKS.index(s.link))


class FlatGroundWorld(BaseWorld):
    """Infinite ground plane z = 0; the fly is free (``world.py:210-331``)."""

    def __init__(self, name: str = "flat_ground_world", *, half_size: float = 1000) -> None:
        super().__init__(name)
        self.half_size = float(half_size)

    def _attach_fly(
        self, fly, spawn_position, spawn_rotation, *,
        bodysegs_with_ground_contact=ContactBodiesPreset.LEGS_THORAX_ABDOMEN_HEAD,
        ground_contact_params: ContactParams = ContactParams(),
        add_ground_contact_sensors: bool = True,
    ) -> str:
        if isinstance(bodysegs_with_ground_contact, (ContactBodiesPreset, str)):
            bodysegs_with_ground_contact = ContactBodiesPreset(
                bodysegs_with_ground_contact
            ).to_body_segments_list()
        ground_contact_params.is_valid()
        self.bodysegs_with_ground_contact = list(bodysegs_with_ground_contact)
        self.ground_contact_params = ground_contact_params
        self.add_ground_contact_sensors = bool(add_ground_contact_sensors)
        if add_ground_contact_sensors:
            per_leg = {}
            for seg in self.bodysegs_with_ground_contact:
                if seg.is_leg():
                    per_leg.setdefault(seg.pos, []).append(seg)
            self.legpos_to_groundcontactsensors_by_fly = {
                fly.name: {
                    leg: dict(name=f"ground_contact_{leg}_leg", subtree_root=_sort_prox2dist(segs)[0].name)
                    for leg, segs in per_leg.items()
                }
            }
        return f"{fly.name}/"


class _TerrainWorld(FlatGroundWorld):
    """Ground whose height is a piecewise-constant function of (x, y).

    The reference snapshot has only the flat plane and the tether (flygym 1.x's terrains were
    dropped; its extension point is ``BaseWorld._attach_fly_mjcf``, reference ``world.py:70-93``, where box geoms would
    be added); the terrains below are build-defined.  The ground is a lattice of axis-aligned cells of constant height
    — i.e. boxes: every collision probe (capsule end sphere, hull vertex) is tested against the TOP of the cell it is
    over (vertical normal) and against the SIDE FACES of the cells around it (horizontal normals), see
    :func:`terrain_probe`.  Round 2 had the tops only (a height field: a tarsus swung into the side of a block was thrown
    upwards by the block's top once inside it).
    """

    def terrain_height(self, x, y):
        """numpy restatement of the height function shared by the oracle and the HIP kernel."""
        import numpy as np

        return _terrain_height(self.terrain_type, self.terrain_params, np.asarray(x, dtype=float), np.asarray(y, dtype=float))

    def terrain_probe(self, p, rho=0.0):
        """numpy restatement of the probe-vs-terrain rule shared by the oracle and the HIP kernel: :func:`terrain_probe`."""
        return terrain_probe(self.terrain_type, self.terrain_params, p, rho)


def _terrain_height(kind, p, x, y):
    import numpy as np

    if kind == 1:      # gapped: blocks of width p0 separated by gaps of width p1 and depth p2, perpendicular to x
        u = x - np.floor(x / (p[0] + p[1])) * (p[0] + p[1])
        return np.where(u < p[0], 0.0, -p[2])
    if kind == 2:      # blocks: checkerboard of squares of side p0, every other square raised by p1
        i, j = np.floor(x / p[0]), np.floor(y / p[0])
        return np.where(np.mod(i + j, 2.0) != 0, p[1], 0.0)
    if kind == 3:      # mixed: stripes of length p3 along x cycling flat -> gapped -> blocks
        stripe = np.mod(np.floor(x / p[3]), 3.0)
        gap = _terrain_height(1, (1.0, p[1], p[2], 0.0), x, y)
        blk = _terrain_height(2, (p[0], 0.35, 0.0, 0.0), x, y)
        return np.where(stripe == 1, gap, np.where(stripe == 2, blk, 0.0))
    return np.zeros(np.broadcast(x, y).shape)


WALL_NORMALS = ((1.0, 0.0, 0.0), (-1.0, 0.0, 0.0), (0.0, 1.0, 0.0), (0.0, -1.0, 0.0))     # wall codes 1..4
_PROBE_EPS = 1e-4     # how far across a cell boundary the neighbour's height is read (cells are >= 0.3 mm wide)


def _cell_bounds(kind, p, x, y):
    """Bounds (x_lo, x_hi, y_lo, y_hi) of the constant-height cell of the lattice that holds (x, y); +-inf where the
    lattice does not divide that axis."""
    import numpy as np

    inf = np.inf
    if kind == 1:
        period = p[0] + p[1]
        k = np.floor(x / period)
        u = x - k * period
        return (k * period if u < p[0] else k * period + p[0]), (k * period + p[0] if u < p[0] else (k + 1) * period), -inf, inf
    if kind == 2:
        i, j = np.floor(x / p[0]), np.floor(y / p[0])
        return i * p[0], (i + 1) * p[0], j * p[0], (j + 1) * p[0]
    if kind == 3:
        st = np.floor(x / p[3])
        k = st - 3 * np.floor(st / 3)
        lo, hi = st * p[3], (st + 1) * p[3]
        if k == 1:
            a, b, c, d = _cell_bounds(1, (1.0, p[1], p[2], 0.0), x, y)
        elif k == 2:
            a, b, c, d = _cell_bounds(2, (p[0], 0.35, 0.0, 0.0), x, y)
        else:
            a, b, c, d = -inf, inf, -inf, inf
        return max(a, lo), min(b, hi), c, d
    return -inf, inf, -inf, inf


def terrain_probe(kind, params, p, rho=0.0):
    """One collision probe — a point ``p`` (hull vertex, ``rho`` = 0) or a sphere of radius ``rho`` centred at ``p``
    (capsule end) — against a terrain of box cells.  Returns ``(dist_top, dist_wall, wall)``:

    * ``dist_top``: signed distance of the probe's lowest point to the top of the cell it is over (normal +z), or +inf
      when the probe is inside that cell's box and its nearest way out is through a side face;
    * ``dist_wall`` / ``wall``: signed distance to the nearest side face that faces the probe and the face's code
      (1..4 = outward normal +x, -x, +y, -y; 0 and +inf if there is none).

    With ``z_b = p_z - rho`` the probe's lowest point, ``h0`` the height of its cell, ``delta_e`` the distance from ``p``
    to the cell's boundary in direction e and ``h_e`` the height of the cell across it:

    * a neighbour with ``h_e > z_b`` concerns the probe.  With the centre below the neighbour's top (``p_z < h_e``; always
      so for a point) it shows a side face at ``dist = delta_e - rho`` whose outward normal is -e.  With the centre above
      it by ``v = p_z - h_e`` (only possible for a sphere, ``0 <= v < rho``) the sphere reaches over the neighbour's top
      EDGE, and the box's nearest feature is axis-aligned only approximately: nearer the face than the top
      (``delta_e >= v``) it is the face as above, otherwise the neighbour's top carries it, ``dist = z_b - h_e`` with
      normal +z (round 4: before, the face applied for every ``h_e > z_b``, so a sphere rolling off a cell's top edge went
      from a 1 um top contact to a ``rho``-deep wall contact; now the depth is continuous across the edge, and across
      the 45 degree line ``delta_e = v`` where the normal turns);
    * ``z_b >= h0`` (above its own cell): ``dist_top = z_b - h0``, or the distance to a neighbour's top found above if
      that is smaller;
    * ``z_b < h0`` (inside its own cell's box): the ways out are up (``h0 - z_b``) and sideways through every face with a
      neighbour the probe would be clear of (``h_e <= z_b``: ``delta_e + rho``).  The shortest wins: sideways gives a face
      at ``dist = -(delta_e + rho)`` with outward normal +e and ``dist_top`` = the nearest neighbour's top (+inf if none);
      up gives ``dist_top = z_b - h0``;
    * of all the faces found the one with the smallest distance is reported.

    A contact exists where a distance is within the pair's margin; its point is the probe's surface point along the
    normal moved half the distance back (as for the plane).  Neighbour heights are read ``1e-4`` mm across the boundary.
    """
    import numpy as np

    x, y, z = (float(v) for v in p)
    h0 = float(_terrain_height(kind, params, np.float64(x), np.float64(y)))
    x_lo, x_hi, y_lo, y_hi = _cell_bounds(kind, params, x, y)
    zb = z - rho
    delta = (x_hi - x, x - x_lo, y_hi - y, y - y_lo)                     # towards +x, -x, +y, -y
    across = ((x_hi + _PROBE_EPS, y), (x_lo - _PROBE_EPS, y), (x, y_hi + _PROBE_EPS), (x, y_lo - _PROBE_EPS))
    he = [float(_terrain_height(kind, params, np.float64(a), np.float64(b))) if np.isfinite(d) else h0
          for d, (a, b) in zip(delta, across)]
    best, code, edge_top = np.inf, 0, np.inf
    for e in range(4):                                                    # neighbours that reach above the probe's lowest point
        if not (np.isfinite(delta[e]) and he[e] > zb):
            continue
        if z - he[e] > delta[e]:                                          # over the neighbour's top edge, nearer its top
            edge_top = min(edge_top, zb - he[e])
        elif delta[e] - rho < best:                                       # its side face
            best, code = delta[e] - rho, (2, 1, 4, 3)[e]                  # normal -e
    if zb >= h0:
        return min(zb - h0, edge_top), best, code
    pen, out = h0 - zb, 0
    for e in range(4):                                                    # inside its own cell's box: the ways out
        if np.isfinite(delta[e]) and he[e] <= zb and delta[e] + rho < pen:
            pen, out = delta[e] + rho, (1, 2, 3, 4)[e]                    # through the face towards e: normal +e
    if out == 0:
        return min(zb - h0, edge_top), best, code
    return (edge_top, -pen, out) if -pen < best else (edge_top, best, code)


class GappedTerrainWorld(_TerrainWorld):
    """Blocks ``block_width`` mm wide separated by gaps ``gap_width`` mm wide and ``gap_depth`` mm deep, across x."""

    def __init__(self, name: str = "gapped_terrain_world", *, block_width: float = 1.0, gap_width: float = 0.3,
                 gap_depth: float = 2.0, half_size: float = 1000) -> None:
        super().__init__(name, half_size=half_size)
        self.terrain_type, self.terrain_params = 1, (float(block_width), float(gap_width), float(gap_depth), 0.0)


class BlocksTerrainWorld(_TerrainWorld):
    """Checkerboard of ``block_size`` mm squares, alternate squares raised by ``height`` mm."""

    def __init__(self, name: str = "blocks_terrain_world", *, block_size: float = 1.3, height: float = 0.35,
                 half_size: float = 1000) -> None:
        super().__init__(name, half_size=half_size)
        self.terrain_type, self.terrain_params = 2, (float(block_size), float(height), 0.0, 0.0)


class MixedTerrainWorld(_TerrainWorld):
    """Stripes ``stripe_length`` mm long along x cycling flat, gapped (1.0 / gap_width / gap_depth) and blocks
    (block_size, 0.35 mm)."""

    def __init__(self, name: str = "mixed_terrain_world", *, block_size: float = 1.3, gap_width: float = 0.3,
                 gap_depth: float = 2.0, stripe_length: float = 4.0, half_size: float = 1000) -> None:
        super().__init__(name, half_size=half_size)
        self.terrain_type = 3
        self.terrain_params = (float(block_size), float(gap_width), float(gap_depth), float(stripe_length))


class TetheredWorld(BaseWorld):
    """Fly body held in space (``world.py:334-366``): the thorax is welded to its spawn pose by a stiff soft
    constraint (six bilateral rows), as in the reference."""

    def __init__(self, name: str = "tethered_world") -> None:
        super().__init__(name)
        self.fixed_base = True

    def _attach_fly(self, fly, spawn_position, spawn_rotation) -> str:
        return f"{fly.name}/"
