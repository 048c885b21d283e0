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

__all__ = ["BaseWorld", "FlatGroundWorld", "TetheredWorld"]


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
        self._compiled = None

    @property
    def fly_lookup(self) -> dict[str, Fly]:
        return self._fly_lookup

    @abstractmethod
    def _attach_fly(self, fly: Fly, spawn_position, spawn_rotation: Rotation3D, *args, **kwargs) -> str:
        """Record how the fly hangs in this world; returns the free-joint name."""

    def add_fly(self, fly: Fly, spawn_position, spawn_rotation: Rotation3D, *args: Any, **kwargs: Any) -> None:
        if fly.name in self._fly_lookup:
            raise ValueError(f"Fly with name '{fly.name}' already exists in the world.")
        if self._fly_lookup:
            raise NotImplementedError(
                "one fly per world: the batch axis of HIPSimulation is the way to run many flies"
            )
        self._fly_lookup[fly.name] = fly
        freejoint = self._attach_fly(fly, spawn_position, spawn_rotation, *args, **kwargs)
        if spawn_rotation.format != "quat":
            raise ValueError(
                "Freejoint neutral rotation can only be specified in quaternion format "
                f"for now. Got {spawn_rotation}."
            )
        self.spawn_position = np.asarray(spawn_position, dtype=np.float64)
        self.spawn_quat = spawn_rotation.as_quat()
        self.world_dof_neutral_states[freejoint] = [*self.spawn_position, *spawn_rotation.values]
        self._compiled = None

    def compile(self):
        from ..compiler.model import compile_world

        if self._compiled is None:
            self._compiled = compile_world(self)
        return self._compiled


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


class TetheredWorld(BaseWorld):
    """Fly body held in space (``world.py:334-366``).  The reference welds the thorax to the
    world with a stiff soft constraint; the engine fixes the base kinematically instead
    (root pose constant, root dofs removed from the solve) — see DESIGN.md."""

    def __init__(self, name: str = "tethered_world") -> None:
        super().__init__(name)
        self.fixed_base = True

    def _attach_fly(self, fly, spawn_position, spawn_rotation) -> str:
        return f"{fly.name}/"
