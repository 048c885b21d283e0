"""Exceptions (mirror of reference ``src/flygym/utils/exceptions.py``)."""


class FlyGymInternalError(Exception):
    """An invariant of the package itself was violated (a bug, not a user error)."""
