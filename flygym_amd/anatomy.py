"""Fly anatomy vocabulary for the MI355X stepping engine.

This is the host-side mirror of the reference's segment / joint / DoF naming
(reference: ``src/flygym/anatomy.py:176-224`` for the segment tables,
``:388-460`` joint presets, ``:463-498`` actuated-DoF presets, ``:501-562``
contact presets, ``:565-634`` ``Skeleton``).  Every array that crosses the
drop-in boundary (``HIPSimulation.get_joint_angles`` etc.) is ordered by the
orders defined here, so the *orders* must equal the reference's; the
implementation is table driven and independent.

Naming: a segment is ``"{pos}_{link}"`` with ``pos`` in ``c, l, r, lf, lm, lh,
rf, rm, rh``; a DoF is ``"{parent}-{child}-{axis}"`` with axis in
``pitch, roll, yaw``.
"""

from __future__ import annotations

from dataclasses import dataclass, field
from enum import Enum
from typing import Iterable, Iterator

__all__ = [
    "RotationAxis", "AxesSet", "AxisOrder", "JointPreset", "ActuatedDOFPreset",
    "ContactBodiesPreset", "BodySegment", "JointDOF", "AnatomicalJoint",
    "Skeleton", "SIDES", "LEGS", "BODY_POSITIONS", "LEG_LINKS", "ANTENNA_LINKS",
    "PROBOSCIS_LINKS", "ABDOMEN_LINKS", "PASSIVE_TARSAL_LINKS",
    "ALL_CONNECTED_SEGMENT_PAIRS", "ALL_SEGMENT_NAMES",
]

# --------------------------------------------------------------------------
# segment tables
# --------------------------------------------------------------------------
SIDES = ["l", "r"]
LEGS = [s + p for s in SIDES for p in "fmh"]  # lf lm lh rf rm rh
BODY_POSITIONS = ["c"] + SIDES + LEGS

LEG_LINKS = ["coxa", "trochanterfemur", "tibia"] + ["tarsus%d" % i for i in range(1, 6)]
ANTENNA_LINKS = ["pedicel", "funiculus", "arista"]
PROBOSCIS_LINKS = ["rostrum", "haustellum"]
ABDOMEN_LINKS = ["abdomen12", "abdomen3", "abdomen4", "abdomen5", "abdomen6"]
PASSIVE_TARSAL_LINKS = ["tarsus%d" % i for i in range(2, 6)]


def _chain(root: str, prefix: str, links: list[str]) -> list[tuple[str, str]]:
    names = [root] + ["%s_%s" % (prefix, lk) for lk in links]
    return list(zip(names[:-1], names[1:]))


def _build_pairs() -> list[tuple[str, str]]:
    pairs = [("c_thorax", "c_head")]
    pairs += _chain("c_head", "c", PROBOSCIS_LINKS)
    pairs += _chain("c_thorax", "c", ABDOMEN_LINKS)
    pairs += [("c_head", s + "_eye") for s in SIDES]
    for s in SIDES:
        pairs += _chain("c_head", s, ANTENNA_LINKS)
    pairs += [("c_thorax", s + "_wing") for s in SIDES]
    pairs += [("c_thorax", s + "_haltere") for s in SIDES]
    for leg in LEGS:
        pairs += _chain("c_thorax", leg, LEG_LINKS)
    return pairs


ALL_CONNECTED_SEGMENT_PAIRS = _build_pairs()
ALL_SEGMENT_NAMES = list(dict.fromkeys(n for pr in ALL_CONNECTED_SEGMENT_PAIRS for n in pr))
_SEGMENT_SET = frozenset(ALL_SEGMENT_NAMES)


# --------------------------------------------------------------------------
# axes
# --------------------------------------------------------------------------
class RotationAxis(Enum):
    PITCH = "pitch"
    ROLL = "roll"
    YAW = "yaw"
    P = "pitch"
    R = "roll"
    Y = "yaw"

    @classmethod
    def _missing_(cls, value):
        if isinstance(value, str):
            key = value.strip().lower()
            short = {"p": cls.PITCH, "r": cls.ROLL, "y": cls.YAW}
            if key in short:
                return short[key]
            for member in (cls.PITCH, cls.ROLL, cls.YAW):
                if member.value == key:
                    return member
        return None

    def to_vector(self) -> tuple[float, float, float]:
        """Unit axis in the child-body frame (x=yaw, y=pitch, z=roll)."""
        return _AXIS_VEC[self.value]

    def to_letter_xyz(self) -> str:
        return _AXIS_LETTER[self.value]


_AXIS_VEC = {"pitch": (0, 1, 0), "roll": (0, 0, 1), "yaw": (1, 0, 0)}
_AXIS_LETTER = {"pitch": "y", "roll": "z", "yaw": "x"}


class AxesSet(set):
    """Set of ``RotationAxis`` that coerces strings on insertion."""

    def __init__(self, items: Iterable | None = None):
        super().__init__()
        for it in items or ():
            self.add(it)

    def add(self, value):
        super().add(RotationAxis(value))

    def remove(self, value):
        super().remove(RotationAxis(value))

    def __contains__(self, value):
        try:
            return super().__contains__(RotationAxis(value))
        except ValueError:
            return False


_P, _R, _Y = RotationAxis.PITCH, RotationAxis.ROLL, RotationAxis.YAW


class AxisOrder(Enum):
    """Order in which single-axis hinges are chained inside one anatomical joint."""

    PITCH_ROLL_YAW = (_P, _R, _Y)
    PITCH_YAW_ROLL = (_P, _Y, _R)
    ROLL_PITCH_YAW = (_R, _P, _Y)
    ROLL_YAW_PITCH = (_R, _Y, _P)
    YAW_PITCH_ROLL = (_Y, _P, _R)
    YAW_ROLL_PITCH = (_Y, _R, _P)
    PRY = (_P, _R, _Y)
    PYR = (_P, _Y, _R)
    RPY = (_R, _P, _Y)
    RYP = (_R, _Y, _P)
    YPR = (_Y, _P, _R)
    YRP = (_Y, _R, _P)
    DONTCARE = (_P, _R, _Y)

    @classmethod
    def _missing_(cls, value):
        if isinstance(value, str):
            value = value.split("_")
        if isinstance(value, (list, tuple)) and len(value) == 3:
            key = tuple(RotationAxis(v) for v in value)
            for member in cls:
                if member.value == key:
                    return member
        return None

    def to_letters_xyz(self) -> str:
        return "".join(a.to_letter_xyz() for a in self.value)

    def to_list_of_str(self) -> list[str]:
        return [a.value for a in self.value]

    def to_str(self) -> str:
        return "_".join(self.to_list_of_str())


# --------------------------------------------------------------------------
# segments, DoFs, joints
# --------------------------------------------------------------------------
@dataclass(frozen=True)
class BodySegment:
    name: str

    def __post_init__(self):
        if self.name not in _SEGMENT_SET:
            raise ValueError(
                f"Invalid body segment name: {self.name}. Must be one of {ALL_SEGMENT_NAMES}."
            )

    @property
    def pos(self) -> str:
        return self.name.split("_")[0]

    @property
    def link(self) -> str:
        return self.name.split("_")[1]

    def is_thorax(self) -> bool:
        return self.name == "c_thorax"

    def is_head(self) -> bool:
        return self.name == "c_head"

    def is_proboscis(self) -> bool:
        return self.link in PROBOSCIS_LINKS

    def is_eye(self) -> bool:
        return self.link == "eye"

    def is_antenna(self) -> bool:
        return self.link in ANTENNA_LINKS

    def is_wing(self) -> bool:
        return self.link == "wing"

    def is_haltere(self) -> bool:
        return self.link == "haltere"

    def is_leg(self) -> bool:
        return self.pos in LEGS

    def is_abdomen(self) -> bool:
        return self.link in ABDOMEN_LINKS


def _seg(x) -> BodySegment:
    return x if isinstance(x, BodySegment) else BodySegment(x)


@dataclass(frozen=True)
class JointDOF:
    parent: BodySegment
    child: BodySegment
    axis: RotationAxis

    def __post_init__(self):
        object.__setattr__(self, "parent", _seg(self.parent))
        object.__setattr__(self, "child", _seg(self.child))
        object.__setattr__(self, "axis", RotationAxis(self.axis))

    @property
    def name(self) -> str:
        return f"{self.parent.name}-{self.child.name}-{self.axis.value}"

    @classmethod
    def from_name(cls, name: str) -> "JointDOF":
        try:
            parent, child, axis = name.split("-")
            return cls(BodySegment(parent), BodySegment(child), RotationAxis(axis))
        except Exception as exc:
            raise ValueError(f"Invalid JointDOF name: {name}") from exc


@dataclass
class AnatomicalJoint:
    parent: BodySegment
    child: BodySegment
    axes: AxesSet = field(default_factory=lambda: AxesSet((_P, _R, _Y)))

    def __post_init__(self):
        self.parent = _seg(self.parent)
        self.child = _seg(self.child)
        if not isinstance(self.axes, AxesSet):
            self.axes = AxesSet(self.axes)

    def iter_dofs(self, axis_order: AxisOrder) -> Iterator[JointDOF]:
        for axis in AxisOrder(axis_order).value:
            if axis in self.axes:
                yield JointDOF(self.parent, self.child, axis)

    @property
    def name(self) -> str:
        return f"{self.parent.name}-{self.child.name}"

    def __hash__(self):
        return hash((self.parent, self.child))

    def __eq__(self, other):
        return (
            isinstance(other, AnatomicalJoint)
            and (self.parent, self.child) == (other.parent, other.child)
        )


# --------------------------------------------------------------------------
# presets
# --------------------------------------------------------------------------
def _biological_axes(child: BodySegment) -> tuple[RotationAxis, ...]:
    """Axes present at the joint whose child is ``child`` (reference :423-441)."""
    if child.is_leg():
        if child.link == "coxa":
            return (_P, _R, _Y)
        if child.link == "trochanterfemur":
            return (_P, _R)
        return (_P,)
    return (_P, _R, _Y)


class JointPreset(Enum):
    ALL_POSSIBLE = "all_possible"
    ALL_BIOLOGICAL = "all_biological"
    LEGS_ONLY = "legs_only"
    LEGS_ACTIVE_ONLY = "legs_active_only"

    def to_joint_list(self) -> list[AnatomicalJoint]:
        out = []
        for parent, child in ALL_CONNECTED_SEGMENT_PAIRS:
            c = BodySegment(child)
            if self is JointPreset.ALL_POSSIBLE:
                axes = (_P, _R, _Y)
            else:
                axes = _biological_axes(c)
                if self is not JointPreset.ALL_BIOLOGICAL and not c.is_leg():
                    continue
                if self is JointPreset.LEGS_ACTIVE_ONLY and c.link in PASSIVE_TARSAL_LINKS:
                    continue
            out.append(AnatomicalJoint(BodySegment(parent), c, AxesSet(axes)))
        return out


class ActuatedDOFPreset(Enum):
    ALL = "all"
    LEGS_ONLY = "legs_only"
    LEGS_ACTIVE_ONLY = "legs_active_only"

    def filter(self, jointdofs: list[JointDOF]) -> list[JointDOF]:
        dofs = list(jointdofs)
        if self is ActuatedDOFPreset.ALL:
            return dofs
        dofs = [d for d in dofs if d.child.is_leg()]
        if self is ActuatedDOFPreset.LEGS_ACTIVE_ONLY:
            dofs = [d for d in dofs if d.child.link not in PASSIVE_TARSAL_LINKS]
        return dofs


class ContactBodiesPreset(Enum):
    ALL = "all"
    LEGS_THORAX_ABDOMEN_HEAD = "legs_thorax_abdomen_head"
    LEGS_ONLY = "legs_only"
    TIBIA_TARSUS_ONLY = "tibia_tarsus_only"

    def to_body_segments_list(self) -> list[BodySegment]:
        segs = [BodySegment(n) for n in ALL_SEGMENT_NAMES]
        if self is ContactBodiesPreset.ALL:
            return segs
        if self is ContactBodiesPreset.LEGS_THORAX_ABDOMEN_HEAD:
            return [s for s in segs if s.is_leg() or s.is_thorax() or s.is_abdomen() or s.is_head()]
        segs = [s for s in segs if s.is_leg()]
        if self is ContactBodiesPreset.TIBIA_TARSUS_ONLY:
            segs = [s for s in segs if s.link == "tibia" or s.link.startswith("tarsus")]
        return segs


# --------------------------------------------------------------------------
# skeleton
# --------------------------------------------------------------------------
class Skeleton:
    """Kinematic tree + per-joint DoF sets (reference ``anatomy.py:565-634``).

    ``iter_jointdofs`` walks the tree depth first, children in the order their
    edges were listed, and inside one anatomical joint in ``axis_order``: this is
    the order of ``qpos[7:]`` / ``qvel[6:]`` in the engine.
    """

    def __init__(self, *, axis_order, joint_preset=None, anatomical_joints=None):
        if (joint_preset is None) == (anatomical_joints is None):
            raise ValueError(
                "Skeleton must be initiated from either joint_preset or "
                "anatomical_joints, but not both."
            )
        if joint_preset is not None:
            anatomical_joints = JointPreset(joint_preset).to_joint_list()
        self.anatomical_joints = list(anatomical_joints)
        self.joint_lookup = {(j.parent, j.child): j for j in self.anatomical_joints}
        self.body_segments = list(
            dict.fromkeys(s for edge in self.joint_lookup for s in edge)
        )
        self.axis_order = AxisOrder(axis_order)
        self._children = self._validate_tree()

    def _validate_tree(self) -> dict[BodySegment, list[BodySegment]]:
        nodes = self.body_segments
        edges = list(self.joint_lookup)
        if len({frozenset(e) for e in edges}) != len(edges) or any(a == b for a, b in edges):
            raise ValueError("Skeleton is invalid - must be a tree.")
        adj = {n: [] for n in nodes}
        for a, b in edges:
            adj[a].append(b)
            adj[b].append(a)
        if nodes:
            if len(edges) != len(nodes) - 1:
                raise ValueError("Skeleton is invalid - must be a tree.")
            seen, todo = set(), [nodes[0]]
            while todo:
                n = todo.pop()
                if n not in seen:
                    seen.add(n)
                    todo.extend(adj[n])
            if len(seen) != len(nodes):
                raise ValueError("Skeleton is invalid - must be a tree.")
        return adj

    def get_tree(self):
        """The skeleton as a :class:`flygym_amd.utils.math.Tree` over its body segments (reference ``anatomy.py:607-613``)."""
        from .utils.math import Tree

        try:
            return Tree(nodes=self.body_segments, edges=list(self.joint_lookup))
        except ValueError as e:
            raise ValueError("Skeleton is invalid - must be a tree.") from e

    def iter_edges(self, root="c_thorax") -> Iterator[tuple[BodySegment, BodySegment]]:
        root = _seg(root)
        if root not in self._children:
            raise ValueError(f"Root '{root}' not in tree")
        seen = {root}

        def walk(node):
            for nb in self._children[node]:
                if nb not in seen:
                    seen.add(nb)
                    yield node, nb
                    yield from walk(nb)

        yield from walk(root)

    def iter_jointdofs(self, root="c_thorax") -> Iterator[JointDOF]:
        for parent, child in self.iter_edges(root):
            joint = self.joint_lookup.get((parent, child))
            if joint is None:  # edge stored child->parent relative to this root
                joint = self.joint_lookup[(child, parent)]
            yield from joint.iter_dofs(self.axis_order)

    def get_actuated_dofs_from_preset(self, preset) -> list[JointDOF]:
        return ActuatedDOFPreset(preset).filter(list(self.iter_jointdofs()))
