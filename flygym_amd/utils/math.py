"""``Rotation3D`` and small helpers (mirror of reference ``src/flygym/utils/math.py:114-175``)."""

from dataclasses import dataclass
from numbers import Number
from typing import Sequence

import numpy as np

__all__ = ["Rotation3D", "orderedset"]

_DIMS = {"quat": 4, "axisangle": 4, "xyaxes": 6, "zaxis": 3, "euler": 3}


def orderedset(items):
    return list(dict.fromkeys(items))


@dataclass(frozen=True)
class Rotation3D:
    format: str
    values: Sequence[Number]

    def __post_init__(self):
        ok = (
            self.format in _DIMS
            and isinstance(self.values, Sequence)
            and all(isinstance(v, Number) for v in self.values)
        )
        if not ok:
            raise ValueError(
                f"Invalid rotation spec: format={self.format}, values={self.values}. "
                f"Format must be one of {list(_DIMS)} and values must be a sequence of numbers."
            )
        if len(self.values) != _DIMS[self.format]:
            raise ValueError(
                f"Invalid rotation spec: format={self.format}, values={self.values}. "
                f"Format {self.format} should be {_DIMS[self.format]}-dimensional, got {len(self.values)}."
            )

    def as_kwargs(self):
        return {self.format: self.values}

    def as_quat(self) -> np.ndarray:
        """(w, x, y, z); only the formats the engine can spawn from."""
        if self.format == "quat":
            q = np.asarray(self.values, dtype=np.float64)
            return q / np.linalg.norm(q)
        if self.format == "axisangle":
            ax = np.asarray(self.values[:3], dtype=np.float64)
            ax = ax / np.linalg.norm(ax)
            h = 0.5 * float(self.values[3])
            return np.array([np.cos(h), *(ax * np.sin(h))])
        raise ValueError(f"cannot convert rotation format '{self.format}' to a quaternion")
