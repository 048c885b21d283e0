"""Static joint-angle poses (mirror of reference ``src/flygym/compose/pose.py:14-161``).

Pose data comes from the asset pack (``scripts/build_asset_pack.py`` stores the six
``pose/neutral/<axis_order>.yaml`` documents) or from a YAML path given by the user.
Left-side angles are mirrored to missing right-side DoFs *without* a sign change —
the model's right-side roll/yaw axes are already mirrored (``fly.py:279-283``).
"""

from __future__ import annotations

import json
from enum import Enum
from pathlib import Path

import numpy as np

from ..anatomy import AxisOrder, BodySegment, JointDOF

__all__ = ["KinematicPose", "KinematicPosePreset"]


def _parse_pose_document(doc: dict):
    unit = doc.get("angle_unit")
    if unit not in ("degree", "radian"):
        raise ValueError("YAML file must contain angle_unit: 'degree' or 'radian'.")
    angles = doc.get("joint_angles")
    if not isinstance(angles, dict):
        raise ValueError("YAML file must contain 'joint_angles' mapping.")
    out = {}
    for k, v in angles.items():
        if isinstance(v, bool) or not isinstance(v, (int, float)):
            raise ValueError(f"Joint angle for '{k}' must be a number.")
        out[k] = float(np.deg2rad(v)) if unit == "degree" else float(v)
    raw = doc.get("axis_order")
    try:
        order = AxisOrder(raw)
    except (ValueError, TypeError):
        raise ValueError(f"Invalid or missing axis_order: {raw}")
    return out, order


def _mirror_left_to_right(angles: dict[str, float]) -> None:
    for name, val in list(angles.items()):
        dof = JointDOF.from_name(name)
        if dof.child.name[0] != "l":
            continue
        parent = dof.parent.name
        if parent[0] == "l":
            parent = "r" + parent[1:]
        twin = JointDOF(BodySegment(parent), BodySegment("r" + dof.child.name[1:]), dof.axis)
        angles.setdefault(twin.name, float(val))


class KinematicPose:
    def __init__(self, *, path=None, joint_angles_rad_dict=None, axis_order=None, mirror_left2right=True):
        if joint_angles_rad_dict is not None and path is None:
            if axis_order is None:
                raise ValueError(
                    "When initializing from `joint_angles_rad_dict`, axis_order must also be provided."
                )
            axis_order = AxisOrder(axis_order)
            angles = dict(joint_angles_rad_dict)
        elif path is not None and joint_angles_rad_dict is None:
            if axis_order is not None:
                raise ValueError(
                    "When initializing from `path`, `axis_order` should not be provided "
                    "because it will be loaded from the pose file."
                )
            import yaml

            angles, axis_order = _parse_pose_document(yaml.safe_load(Path(path).read_text()))
        else:
            raise ValueError("Either joint_angles_rad_dict or path must be provided, but not both.")
        if mirror_left2right:
            _mirror_left_to_right(angles)
        self.axis_order = axis_order
        self.joint_angles_lookup_rad = angles

    def copy(self) -> "KinematicPose":
        return KinematicPose(
            joint_angles_rad_dict=dict(self.joint_angles_lookup_rad), axis_order=self.axis_order,
            mirror_left2right=False,
        )


class KinematicPosePreset(Enum):
    NEUTRAL = "neutral"

    def get_dir(self) -> Path:
        """Directory of this preset's pose files, one YAML per axis order (reference ``pose.py:140-145``)."""
        return Path(__file__).resolve().parents[1] / "assets" / "model" / "pose" / self.value

    def get_pose_by_axis_order(self, axis_order, mirror_left2right: bool = True) -> KinematicPose:
        from ..compiler.model import load_asset_pack

        docs = json.loads(str(load_asset_pack()[f"{self.value}_pose_json"]))
        key = AxisOrder(axis_order).to_str()
        if key not in docs:
            raise ValueError(f"no '{self.value}' pose stored for axis order {key}")
        angles, order = _parse_pose_document(docs[key])
        return KinematicPose(joint_angles_rad_dict=angles, axis_order=order, mirror_left2right=mirror_left2right)
