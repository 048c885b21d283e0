"""Contact parameters (host-side mirror of reference ``src/flygym/compose/physics.py:6-165``).

Same field names, defaults, tuple layouts and validation messages as the reference
dataclass so user code and the reference's own ``tests/core/test_physics.py`` cases
carry over.  Two reference behaviours are kept on purpose because they change the
physics the engine must reproduce:

* ``get_solimp_tuple`` returns FOUR numbers (dmin, dmax, midpoint, sharpness) — the
  width is not passed on (``physics.py:103-111``), so the engine sees
  ``solimp = (0.98, 0.99, width=0.5, midpoint=3.0→clamped, power=2)``;
* the default sliding friction is 1.0 (``physics.py:61``).
"""

from dataclasses import dataclass

__all__ = ["ContactParams"]


@dataclass(kw_only=True)
class ContactParams:
    sliding_friction: float = 1.0
    torsional_friction: float = 2e-2
    rolling_friction: float = 1e-4
    solver_refaccl_timeconst: float = 2e-4
    solver_refaccl_dampratio: float = 1.0
    solver_impedance_min: float = 0.98
    solver_impedance_max: float = 0.99
    solver_impedance_min2max_width: float = 1e-5
    solver_impedance_transitionmidpoint: float = 0.5
    solver_impedance_transitionsharpness: float = 3.0
    margin: float = 1e-3

    # tuples in the layout MuJoCo's <contact><pair> expects -----------------------
    def get_friction_tuple(self):
        self._check_friction()
        s, t, r = self.sliding_friction, self.torsional_friction, self.rolling_friction
        return (s, s, t, r, r)

    def get_solref_tuple(self):
        self._check_refaccl()
        return (self.solver_refaccl_timeconst, self.solver_refaccl_dampratio)

    def get_solimp_tuple(self):
        self._check_impedance()
        return (
            self.solver_impedance_min,
            self.solver_impedance_max,
            self.solver_impedance_transitionmidpoint,
            self.solver_impedance_transitionsharpness,
        )

    def is_valid(self, raise_on_invalid: bool = True) -> bool:
        try:
            self._check_friction()
            self._check_refaccl()
            self._check_impedance()
        except ValueError as exc:
            if raise_on_invalid:
                raise ValueError(f"Invalid ContactParams: {exc}") from exc
            return False
        return True

    # validation ------------------------------------------------------------------
    def _check_friction(self):
        for label, v in (("Sliding", self.sliding_friction), ("Torsional", self.torsional_friction),
                         ("Rolling", self.rolling_friction)):
            if not v >= 0:
                raise ValueError(f"{label} friction must be non-negative")

    def _check_refaccl(self):
        if not self.solver_refaccl_timeconst > 0:
            raise ValueError("Solver reference time constant must be positive")
        if not self.solver_refaccl_dampratio > 0:
            raise ValueError("Solver reference damping ratio must be positive")

    def _check_impedance(self):
        if not 0 < self.solver_impedance_min < 1:
            raise ValueError("Minimum solver impedance must be in (0, 1)")
        if not 0 < self.solver_impedance_max < 1:
            raise ValueError("Maximum solver impedance must be in (0, 1)")
        if not self.solver_impedance_max >= self.solver_impedance_min:
            raise ValueError("Maximum solver impedance cannot be less than minimum")
        if not self.solver_impedance_min2max_width > 0:
            raise ValueError("Impedance mid-to-max transition must happen over a positive distance")
        if not 0 < self.solver_impedance_transitionmidpoint < 1:
            raise ValueError("Midpoint of impedance min-to-max must be in (0, 1)")
        if not self.solver_impedance_transitionsharpness >= 1:
            raise ValueError("Sharpness of impedance transition must be at least linear (1)")
