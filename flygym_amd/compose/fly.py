"""``Fly``: the model description the engine compiles (mirror of reference
``src/flygym/compose/fly.py:80-678``).

The reference ``Fly`` edits a ``dm_control.mjcf`` document and recompiles it with
MuJoCo after every change.  This class records the same choices — segments, joints,
actuators, adhesion, sites, collision-geometry fitting — as plain data; the
:mod:`flygym_amd.compiler` turns them into the flat model the HIP kernels read.
Method names, argument names, defaults, orders and error messages follow the
reference so scripts written against it run unchanged.
"""

from __future__ import annotations

import json
from enum import Enum
from typing import Any, Iterable

import numpy as np

from ..anatomy import (
    LEGS, AnatomicalJoint, AxisOrder, BodySegment, JointDOF, JointPreset, RotationAxis, Skeleton,
)
from ..utils.math import Rotation3D
from .pose import KinematicPose, KinematicPosePreset

__all__ = ["Fly", "ActuatorType", "MeshType", "GeomFittingOption"]


class MeshType(Enum):
    FULLSIZE = "fullsize"
    SIMPLIFIED_MAX2000FACES = "simplified_max2000faces"


class GeomFittingOption(Enum):
    UNMODIFIED = "unmodified"
    ALL_TO_CAPSULES = "all_to_capsules"
    CLAWS_TO_CAPSULES = "claws_to_capsules"


class ActuatorType(Enum):
    MOTOR = "motor"
    POSITION = "position"
    VELOCITY = "velocity"
    INTVELOCITY = "intvelocity"
    DAMPER = "damper"
    CYLINDER = "cylinder"
    MUSCLE = "muscle"
    ADHESION = "adhesion"


# Stateless affine actuators: force = gain ctrl + bias_q q + bias_v qd (MuJoCo's position: gain kp, bias (-kp, -kv);
# velocity: gain kv, bias (0, -kv); motor: gain gear) run in the stepping kernel's affine pass.  The other joint actuators of the
# reference's ActuatorType (compose/fly.py:65-77) — intvelocity, cylinder, muscle (an activation state per actuator) and damper
# (gain proportional to the joint velocity) — are MuJoCo's general actuator (compiler/model.py::_general_row; round 6).
_AFFINE_ACTUATORS = (ActuatorType.POSITION, ActuatorType.MOTOR, ActuatorType.VELOCITY)
# attributes of the MJCF shortcuts this engine takes (XMLreference.html#actuator-*); anything else is refused by name
_ACTUATOR_ATTRS = {
    ActuatorType.POSITION: ("kp", "kv"), ActuatorType.VELOCITY: ("kv",), ActuatorType.MOTOR: ("gear",),
    ActuatorType.INTVELOCITY: ("kp", "kv", "actrange", "gear"), ActuatorType.DAMPER: ("kv", "gear"),
    ActuatorType.CYLINDER: ("timeconst", "area", "diameter", "bias", "gear"),
    ActuatorType.MUSCLE: ("timeconst", "tausmooth", "range", "force", "scale", "lmin", "lmax", "vmax", "fpmax", "fvmax", "lengthrange", "gear"),
}
_VECTOR_ATTRS = {"actrange": 2, "bias": 3, "range": 2, "lengthrange": 2}


class Fly:
    def __init__(
        self,
        name: str = "nmf",
        *,
        asset_pack_path=None,
        root_segment: BodySegment | str = "c_thorax",
        mirror_left2right: bool = True,
        mesh_type: MeshType = MeshType.SIMPLIFIED_MAX2000FACES,
        geom_fitting_option: GeomFittingOption = GeomFittingOption.UNMODIFIED,
    ) -> None:
        from ..compiler.model import load_asset_pack

        self._name = name
        self.asset_pack_path = asset_pack_path
        pack = load_asset_pack(asset_pack_path)
        self.mujoco_globals = json.loads(str(pack["mujoco_globals_json"]))
        self.skeleton: Skeleton | None = None
        self.root_segment = BodySegment(root_segment) if isinstance(root_segment, str) else root_segment
        self.mirror_left2right = mirror_left2right
        self.mesh_type = MeshType(mesh_type)
        self.geom_fitting_option = GeomFittingOption(geom_fitting_option)

        self.jointdof_to_neutralangle: dict[JointDOF, float] = {}
        self.jointdof_to_neutralaction_by_type = {ty: {} for ty in ActuatorType}
        self.jointdof_to_actuator_by_type = {ty: {} for ty in ActuatorType}
        self.leg_to_adhesionactuator: dict[str, dict] = {}
        self.anatomicaljoint_to_sites: dict[AnatomicalJoint, dict] = {}
        self.cameraname_to_camera: dict[str, dict] = {}
        self.joint_params: dict[JointDOF, dict] = {}
        self.actuators: list[dict] = []   # in creation order == engine ctrl order
        self.colorized = False

        # body segments in the canonical (DFS over every anatomical joint) order, fly.py:545-582
        full = Skeleton(joint_preset=JointPreset.ALL_POSSIBLE, axis_order=AxisOrder.DONTCARE)
        self._segment_edges = [(p.name, c.name) for p, c in full.iter_edges(self.root_segment)]
        self._bodysegs = [self.root_segment] + [BodySegment(c) for _, c in self._segment_edges]
        # every mesh must exist (fly.py:524-536)
        from ..compiler.model import mesh_for_segment
        for seg in self._bodysegs:
            mesh_for_segment(pack, seg.name, self.mesh_type.value, mirror_left2right)

    # ---- identity / orders -------------------------------------------------------
    @property
    def name(self) -> str:
        return self._name

    def segment_edges(self) -> list[tuple[str, str]]:
        return list(self._segment_edges)

    def segment_is_capsule(self, seg_name: str) -> bool:
        """tarsus5 geoms are always capsules (``fly.py:585-589``)."""
        seg = BodySegment(seg_name)
        return self.geom_fitting_option == GeomFittingOption.ALL_TO_CAPSULES or (
            seg.is_leg() and seg.link == "tarsus5"
        )

    # ---- MJCF surface of the reference (compose/base.py): this engine has no MJCF document; said so, not silently absent ----------
    @property
    def mjcf_root(self):
        raise AttributeError(
            "flygym_amd has no MJCF document (the reference's fly.mjcf_root is a dm_control element tree): the recorded choices are "
            "fly.joint_params / fly.set_joint_params(), fly.actuators, fly.mujoco_globals; world.compile_model() gives the compiled model")

    def save_xml_with_assets(self, *args, **kwargs):
        raise NotImplementedError(
            "flygym_amd compiles the recorded choices to its own flat model, not to MJCF: use world.compile_model().save(path) "
            "(CompiledModel.load(path) reads it back) — there is no XML to export")

    def compile(self):
        """``(model, data)`` of the fly on its own, as the reference's ``Fly.compile`` (``compose/base.py:21-27``): the
        fly floating in empty space.  The reference's standalone fly has no free joint, so the summary counts only the
        hinges (``nq == nv ==`` number of joint dofs); the engine's full model (with the free root it always simulates)
        is ``model.compiled``.  Does not attach the fly to any world."""
        from types import SimpleNamespace

        from ..compiler.model import CompiledData
        from ..utils.math import Rotation3D
        from .world import _FreeSpaceWorld

        world = _FreeSpaceWorld()
        world.add_fly(self, (0.0, 0.0, 0.0), Rotation3D("quat", (1, 0, 0, 0)))
        full = world.compile_model()
        n_hinge = full.nv - 6
        summary = SimpleNamespace(nq=n_hinge, nv=n_hinge, njnt=n_hinge, nu=full.nu, nbody=full.nbody, nsite=full.nsite,
                                  ncam=full.ncam, ngeom=0, compiled=full)
        data = CompiledData(full)
        data.qpos, data.qvel = data.qpos[7:], data.qvel[6:]
        return summary, data

    # ---- the reference's lookup names (fly.py:160-170).  There they hold dm_control MJCF elements; here the same keys map
    # to the plain records this package keeps, so code that counts, iterates or tests membership keeps working.
    @property
    def bodyseg_to_mjcfbody(self) -> dict:
        return {seg: {"name": seg.name} for seg in self._bodysegs}

    @property
    def bodyseg_to_mjcfgeom(self) -> dict:
        return {seg: {"name": seg.name, "mesh": self.mesh_type.value} for seg in self._bodysegs}

    @property
    def jointdof_to_mjcfjoint(self) -> dict:
        return self.joint_params

    @property
    def jointdof_to_mjcfactuator_by_type(self) -> dict:
        return self.jointdof_to_actuator_by_type

    @property
    def anatomicaljoint_to_mjcfsites(self) -> dict:
        return self.anatomicaljoint_to_sites

    @property
    def cameraname_to_mjcfcamera(self) -> dict:
        return self.cameraname_to_camera

    def get_bodysegs_order(self) -> list[BodySegment]:
        return list(self._bodysegs)

    def get_jointdofs_order(self) -> list[JointDOF]:
        return list(self.joint_params.keys())

    def get_actuated_jointdofs_order(self, actuator_type) -> list[JointDOF]:
        return list(self.jointdof_to_actuator_by_type[ActuatorType(actuator_type)].keys())

    def get_legs_order(self) -> list[str]:
        return LEGS

    def get_sites_order(self) -> list[AnatomicalJoint]:
        return list(self.anatomicaljoint_to_sites.keys())

    # ---- joints ------------------------------------------------------------------
    def add_joints(
        self,
        skeleton: Skeleton,
        neutral_pose: KinematicPose | KinematicPosePreset | None = None,
        *,
        stiffness: float = 10.0,
        damping: float = 0.5,
        armature: float = 1e-6,
        **kwargs: Any,
    ) -> dict[JointDOF, dict]:
        if neutral_pose is None:
            lookup = {}
        elif isinstance(neutral_pose, KinematicPose):
            lookup = neutral_pose.joint_angles_lookup_rad
        elif isinstance(neutral_pose, KinematicPosePreset):
            lookup = neutral_pose.get_pose_by_axis_order(skeleton.axis_order).joint_angles_lookup_rad
        else:
            raise ValueError(
                "When specified, `neutral_pose` must be a `KinematicPose` or `KinematicPosePreset`."
            )
        if kwargs:
            raise NotImplementedError(
                f"joint attributes {sorted(kwargs)} are not supported by the MI355X engine yet"
            )
        self.skeleton = skeleton
        out = {}
        for dof in skeleton.iter_jointdofs(self.root_segment):
            ang = float(lookup.get(dof.name, 0.0))
            self.jointdof_to_neutralangle[dof] = ang
            vec = np.array(dof.axis.to_vector(), dtype=np.float64)
            # right-side roll / yaw axes are mirrored so that angles are symmetric (fly.py:279-283)
            if dof.child.pos[0] == "r" and dof.axis != RotationAxis.PITCH:
                vec = -vec
            out[dof] = dict(
                name=dof.name, axis=vec, stiffness=float(stiffness), damping=float(damping),
                armature=float(armature), springref=ang,
            )
        self.joint_params.update(out)
        return out

    def set_joint_params(self, jointdofs: Iterable[JointDOF], *, stiffness: float | None = None, damping: float | None = None,
                         armature: float | None = None, springref: float | None = None) -> None:
        """Change the passive parameters of SOME joints after :meth:`add_joints` (which sets one value for all).  The reference does
        this by editing the MJCF tree — ``for joint in fly.mjcf_root.worldbody.find_all("joint"): joint.stiffness = 5``
        (tutorial 1bis, "change the stiffness of all tarsal joints"); this engine has no MJCF document, the recorded joint
        parameters (``fly.joint_params[dof]``: name, axis, stiffness, damping, armature, springref) are what the compiler reads.
        Global options are ``fly.mujoco_globals`` (``["option"]["timestep"]``, ``["option"]["gravity"]``, ...: the reference's
        ``fly.mjcf_root.option``)."""
        for dof in jointdofs:
            if dof not in self.joint_params:
                raise ValueError(f"joint {dof.name} does not exist")
            for key, val in (("stiffness", stiffness), ("damping", damping), ("armature", armature), ("springref", springref)):
                if val is not None:
                    if key != "springref" and float(val) < 0:
                        raise ValueError(f"{key} cannot be negative")
                    self.joint_params[dof][key] = float(val)
            if springref is not None:
                self.jointdof_to_neutralangle[dof] = float(springref)

    # ---- actuators ---------------------------------------------------------------
    def add_actuators(
        self,
        jointdofs: Iterable[JointDOF],
        actuator_type,
        neutral_input=None,
        *,
        forcelimited: bool = True,
        forcerange: tuple[float, float] = (-30.0, 30.0),
        **kwargs: Any,
    ) -> dict[JointDOF, dict]:
        actuator_type = ActuatorType(actuator_type)
        if actuator_type == ActuatorType.ADHESION:
            raise ValueError("adhesion actuators act on body segments: use add_leg_adhesion()")
        if neutral_input is None:
            neutral_input = {}
        if actuator_type == ActuatorType.POSITION:
            if isinstance(neutral_input, KinematicPose):
                neutral_input = neutral_input.joint_angles_lookup_rad
            elif isinstance(neutral_input, KinematicPosePreset):
                neutral_input = neutral_input.get_pose_by_axis_order(
                    self.skeleton.axis_order
                ).joint_angles_lookup_rad
        ctrlrange = kwargs.pop("ctrlrange", None)
        known = {}
        for k in _ACTUATOR_ATTRS[actuator_type]:
            if k not in kwargs:
                continue
            v = kwargs.pop(k)
            if k == "timeconst" and actuator_type == ActuatorType.MUSCLE:
                known[k] = tuple(map(float, v))
                if len(known[k]) != 2:
                    raise ValueError("muscle timeconst takes (activation, deactivation) time constants")
            elif k in _VECTOR_ATTRS:
                known[k] = tuple(map(float, v))
                if len(known[k]) != _VECTOR_ATTRS[k]:
                    raise ValueError(f"actuator attribute {k} takes {_VECTOR_ATTRS[k]} numbers")
            else:
                known[k] = float(v)
        if kwargs:
            raise NotImplementedError(
                f"actuator attributes {sorted(kwargs)} are not supported by the MI355X engine yet"
            )
        # what MuJoCo's compiler checks for these shortcuts (same conditions, its wording where remembered)
        if actuator_type == ActuatorType.INTVELOCITY:
            if "actrange" not in known or not known["actrange"][0] < known["actrange"][1]:
                raise ValueError("intvelocity actuators need actrange=(low, high) with low < high (their activation is clamped to it)")
        if actuator_type == ActuatorType.DAMPER:
            if known.get("kv", 1.0) < 0:
                raise ValueError("damping coefficient cannot be negative")
            if ctrlrange is None or min(ctrlrange) < 0:
                raise ValueError("damper actuators need a non-negative ctrlrange (damper control range cannot be negative)")
        if actuator_type == ActuatorType.CYLINDER and "diameter" in known:
            known["area"] = float(np.pi / 4.0 * known.pop("diameter") ** 2)
        if actuator_type == ActuatorType.MUSCLE:
            if "lengthrange" not in known or not known["lengthrange"][0] < known["lengthrange"][1]:
                raise NotImplementedError(
                    "muscle actuators need lengthrange=(low, high) here: MuJoCo derives it by simulation at compile time "
                    "(mj_setLengthRange) for joints without limits, which this engine does not reproduce")
        out = {}
        for dof in jointdofs:
            if dof not in self.joint_params:
                raise ValueError(f"cannot actuate {dof.name}: joint does not exist")
            neutral = float(neutral_input.get(dof.name, 0.0))
            self.jointdof_to_neutralaction_by_type[actuator_type][dof] = neutral
            act = dict(
                kind=actuator_type.value, name=f"{dof.name}-{actuator_type.value}", jointdof=dof,
                forcelimited=bool(forcelimited), forcerange=tuple(map(float, forcerange)),
                ctrllimited=ctrlrange is not None,
                ctrlrange=tuple(map(float, ctrlrange)) if ctrlrange is not None else (0.0, 0.0),
                neutral=neutral, **known,
            )
            self.actuators.append(act)
            out[dof] = act
        self.jointdof_to_actuator_by_type[actuator_type].update(out)
        return out

    def add_joint_sites(self, anatomical_joints: list[AnatomicalJoint]) -> dict[AnatomicalJoint, dict]:
        out = {}
        for joint in anatomical_joints:
            if joint in self.anatomicaljoint_to_sites:
                raise ValueError(
                    f"A site has already been added for anatomical joint '{joint.name}'."
                )
            out[joint] = dict(name=joint.name, segment=joint.child.name, pos=(0.0, 0.0, 0.0))
            self.anatomicaljoint_to_sites[joint] = out[joint]
        return out

    def add_leg_adhesion(self, gain: float | dict[str, float] = 1.0) -> dict[str, dict]:
        """Adhesion actuators on the six tarsus5 segments; ``ctrlrange=(1, 100)`` with
        autolimits, so a control of 0 still applies ``gain·1`` (``fly.py:407-441``)."""
        if self.leg_to_adhesionactuator:
            raise ValueError("Leg adhesion actuators have already been added.")
        for leg in LEGS:
            g = gain[leg] if isinstance(gain, dict) else gain
            act = dict(
                kind="adhesion", name=f"{leg}_tarsus5-adhesion", segment=f"{leg}_tarsus5",
                gain=float(g), forcelimited=False, forcerange=(0.0, 0.0),
                ctrllimited=True, ctrlrange=(1.0, 100.0), neutral=0.0,
            )
            self.actuators.append(act)
            self.leg_to_adhesionactuator[leg] = act
        return self.leg_to_adhesionactuator

    # ---- visual-only API (kept so reference scripts run; no effect on physics) -----
    def colorize(self, visuals_config_path=None) -> None:
        self.colorized = True

    def add_tracking_camera(
        self, name: str = "trackcam", mode: str = "track", pos_offset=(0, -7.5, 6),
        rotation: Rotation3D = Rotation3D("xyaxes", (1, 0, 0, 0, 0.6, 0.8)), fovy: float = 30.0, **kwargs,
    ) -> dict:
        cam = dict(name=name, mode=mode, target=self.root_segment.name, pos=tuple(pos_offset),
                   rotation=rotation, fovy=float(fovy), **kwargs)
        self.cameraname_to_camera[name] = cam
        return cam
