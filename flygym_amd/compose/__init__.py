from .fly import ActuatorType, Fly, GeomFittingOption, MeshType
from .physics import ContactParams
from .pose import KinematicPose, KinematicPosePreset
from .world import (BaseWorld, BlocksTerrainWorld, FlatGroundWorld, GappedTerrainWorld, MixedTerrainWorld,
                    TetheredWorld)

__all__ = [
    "Fly", "ActuatorType", "MeshType", "GeomFittingOption", "BaseWorld", "FlatGroundWorld",
    "TetheredWorld", "GappedTerrainWorld", "BlocksTerrainWorld", "MixedTerrainWorld", "KinematicPose", "KinematicPosePreset", "ContactParams",
]
