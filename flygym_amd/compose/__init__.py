from .fly import ActuatorType, Fly, GeomFittingOption, MeshType
from .physics import ContactParams
from .pose import KinematicPose, KinematicPosePreset
from .world import BaseWorld, FlatGroundWorld, TetheredWorld

__all__ = [
    "Fly", "ActuatorType", "MeshType", "GeomFittingOption", "BaseWorld", "FlatGroundWorld",
    "TetheredWorld", "KinematicPose", "KinematicPosePreset", "ContactParams",
]
