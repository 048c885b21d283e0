"""flygym_amd — MI355X-native batched NeuroMechFly stepping engine.

Drop-in for the physics hot path of NeLy-EPFL/flygym (``Simulation`` / ``GPUSimulation`` ``step()``):
hand-written HIP kernels for gfx950 behind the reference's Python surface.
"""

from pathlib import Path

from . import anatomy, compose
from .models import make_model

assets_dir = Path(__file__).resolve().parent / "assets"     # as `flygym.assets_dir` (pose files; the asset pack)

__all__ = ["anatomy", "compose", "make_model", "HIPSimulation", "Simulation"]
__version__ = "0.1.0"


def __getattr__(name):
    if name == "HIPSimulation":
        from .simulation import HIPSimulation

        return HIPSimulation
    if name == "Simulation":
        from .simulation import Simulation

        return Simulation
    raise AttributeError(name)
