"""Hand-built one-body models for closed-form contact tests (test helper).

A single free body of mass ``mass`` carrying one contact geom — a sphere, written as a capsule with coincident ends, so
the collision stage reports its two end-sphere contacts at the same point — over a ground plane with an arbitrary unit
normal.  Everything the engine needs is written out by hand in the compiled-model format
(``flygym_amd.compiler.model.CompiledModel``): nothing of the fly's model compiler is involved, so the contact
constants under test (``pair_solref`` / ``pair_solimp`` / friction / margin, ``geom_invweight0``) are exactly the ones
given here.  The rotational inertia is made huge: friction at the contact point cannot spin the body up during a test,
and the translational response at the contact point is 1 / mass.
"""

from __future__ import annotations

import numpy as np

from flygym_amd.compiler.model import CompiledModel, EngineSemantics


def sphere_on_plane(mass=1e-3, radius=0.1, normal=(0.0, 0.0, 1.0), mu=1.0, solref=(2e-4, 1.0),
                    solimp=(0.98, 0.99, 0.5, 0.9999, 2.0), margin=1e-3, gravity=(0.0, 0.0, -9810.0), timestep=1e-4,
                    start_height=None, semantics: EngineSemantics | None = None, terrain=None, start_xy=None,
                    noslip_iterations=0) -> CompiledModel:
    """``terrain``: ``(type, (p0, p1, p2, p3))`` of a build-defined terrain (flygym_amd/compose/world.py) over the z = 0
    plane; ``start_xy``: where the sphere starts (with ``start_height`` above z = 0)."""
    n = np.asarray(normal, dtype=np.float64)
    n = n / np.linalg.norm(n)
    sem = semantics or EngineSemantics()
    m = CompiledModel()
    f, i = (lambda *a: np.asarray(a, dtype=np.float64)), (lambda *a: np.asarray(a, dtype=np.int32))
    big = 1e9 * mass                                    # rotational inertia: effectively no rotation (friction torque ~ m g radius)
    start = (radius if start_height is None else start_height) * n
    if start_xy is not None:
        start = np.array([start_xy[0], start_xy[1], radius if start_height is None else start_height], dtype=np.float64)
    t_type, t_par = (0, (0.0, 0.0, 0.0, 0.0)) if terrain is None else terrain
    t_max = {0: 0.0, 1: 0.0, 2: t_par[1], 3: 0.35}[t_type]
    m.update(
        body_parent=i(-1), body_dofadr=i(0), body_dofnum=i(6), body_pos=f([0, 0, 0]), body_quat=f([1, 0, 0, 0]),
        body_mass=f(mass), body_ipos=f([0, 0, 0]), body_inertia=f([big, big, big, 0, 0, 0]),
        dof_body=i(0, 0, 0, 0, 0, 0), dof_parent=i(-1, 0, 1, 2, 3, 4), dof_axis=np.zeros((6, 3)),
        dof_armature=np.zeros(6), dof_damping=np.zeros(6), dof_stiffness=np.zeros(6), dof_springref=np.zeros(6),
        seg_body=i(0), seg_pos=f([0, 0, 0]), seg_quat=f([1, 0, 0, 0]), seg_invweight0=f([1.0 / mass, 1.0 / big]),
        site_body=np.zeros(0, np.int32), site_pos=np.zeros((0, 3)),
        act_type=np.zeros(0, np.int32), act_trn=np.zeros(0, np.int32), act_limited=np.zeros((0, 2), np.int32),
        act_geom=np.zeros(0, np.int32), act_gain=np.zeros(0), act_bias=np.zeros((0, 2)), act_forcerange=np.zeros((0, 2)),
        act_ctrlrange=np.zeros((0, 2)), key_ctrl=np.zeros(0),
        key_qpos=f(*start, 1, 0, 0, 0), qpos0=f(*start, 1, 0, 0, 0),
        geom_body=i(0), geom_type=i(0), geom_hulladr=i(0), geom_hullnum=i(0), geom_sensor=i(-1), geom_seg=i(0),
        geom_p0=f([0, 0, 0]), geom_p1=f([0, 0, 0]), geom_radius=f(radius), geom_bsphere=f([0, 0, 0, radius]),
        geom_invweight0=f(1.0 / mass), hull_vert=np.zeros((1, 3)), hull_skin=f(1e-3),
        pair_friction=f([mu, mu, 0.02, 1e-4, 1e-4]), pair_solref=f(list(solref)), pair_solimp=f(list(solimp)), pair_margin=f(margin),
        opt_timestep=f(timestep), opt_gravity=f(*gravity), opt_tolerance=f(1e-8), opt_solver=i(100, noslip_iterations),
        stat_meaninertia=f(mass), plane=f(*n, 0.0), terrain_type=i(t_type), terrain_params=f(*t_par, t_max),
        weld_active=i(0), weld_params=np.zeros(16), n_sensor=i(0), star=i(0, 0, 0, 0), sem_options=sem.flags(),
    )
    return m


def impedance(r, solimp):
    """MuJoCo's documented solimp curve d(r): d0, dmax, width, midpoint, power."""
    d0, dmax, width, mid, power = solimp
    x = min(1.0, abs(r) / width)
    if x <= mid:
        y = x ** power / mid ** (power - 1.0)
    else:
        y = 1.0 - (1.0 - x) ** power / (1.0 - mid) ** (power - 1.0)
    return d0 + y * (dmax - d0)


def contact_row_constants(r, mass, mu, solref, solimp, timestep, pyramid_plain=False):
    """(K d(r), B, D) of one pyramid row of a contact at signed position r = distance - margin, from the documented
    formulas with the reference's parameters: K = 1 / (dmax^2 tc^2 zeta^2), B = 2 / (dmax tc) with tc >= 2 dt;
    R = (1 - d) / d * (1 + mu^2) / mass per contact, pyramid rows 2 mu^2 R (or R: pyramid_R = "plain")."""
    tc, zeta = max(solref[0], 2.0 * timestep), solref[1]
    dmax = solimp[1]
    K, B = 1.0 / (dmax * dmax * tc * tc * zeta * zeta), 2.0 / (dmax * tc)
    d = impedance(r, solimp)
    Rn = (1.0 - d) / d * (1.0 + mu * mu) / mass
    R = Rn if pyramid_plain else 2.0 * mu * mu * Rn
    return K * d, B, 1.0 / R


class _NoFly:
    """What HIPSimulation asks of a fly when it builds its index maps: nothing to index here."""
    name = "body"
    actuators: list = []

    def get_jointdofs_order(self):
        return []


class TinyWorld:
    """The part of a world ``HIPSimulation`` reads, around a hand-built compiled model."""

    def __init__(self, model: CompiledModel):
        self._model = model
        self.fly_lookup = {"body": _NoFly()}
        self.noslip_iterations = 0
        self.legpos_to_groundcontactsensors_by_fly = None

    def compile_model(self):
        return self._model


def hinge_on_heavy_base(inertia_yy=2e-6, mass=1e-3, com=(0.5, 0.0, 0.0), armature=1e-6, damping=2e-5, stiffness=0.0, springref=0.0,
                        kp=0.0, kv=0.0, forcerange=None, q0=0.0, timestep=1e-4, servo="position", general=None, ctrlrange=None) -> CompiledModel:
    """A link on a hinge (axis y through the base's origin) carried by a free base 1e9 times heavier, no gravity, no
    contact (the base's only geom floats 100 mm over the plane): the base stays put to 1e-9, so the hinge obeys the
    one-dof equation  (I + armature) qdd = tau_act - stiffness (q - springref) - damping qd  with
    I = inertia_yy + mass |com|^2 (com perpendicular to the axis).  One position actuator on the hinge: force =
    kp ctrl - kp q - kv qd, clamped to ``forcerange`` when given (``servo="velocity"``: MuJoCo's velocity servo,
    force = kv (ctrl - qd))."""
    base_mass = 1e9 * mass
    m = sphere_on_plane(mass=base_mass, radius=0.1, gravity=(0.0, 0.0, 0.0), timestep=timestep, start_height=100.0)
    f, i = (lambda *a: np.asarray(a, dtype=np.float64)), (lambda *a: np.asarray(a, dtype=np.int32))
    big = 1e9 * (inertia_yy + mass * float(np.dot(com, com)))
    limited = 1 if forcerange is not None else 0
    lo, hi = forcerange if forcerange is not None else (0.0, 0.0)
    m.update(
        body_parent=i(-1, 0), body_dofadr=i(0, 6), body_dofnum=i(6, 1), body_pos=f([0, 0, 0], [0, 0, 0]),
        body_quat=f([1, 0, 0, 0], [1, 0, 0, 0]), body_mass=f(base_mass, mass), body_ipos=f([0, 0, 0], list(com)),
        body_inertia=f([big, big, big, 0, 0, 0], [inertia_yy, inertia_yy, inertia_yy, 0, 0, 0]),
        dof_body=i(0, 0, 0, 0, 0, 0, 1), dof_parent=i(-1, 0, 1, 2, 3, 4, 5), dof_axis=f(*([[0, 0, 0]] * 6), [0, 1, 0]),
        dof_armature=f(0, 0, 0, 0, 0, 0, armature), dof_damping=f(0, 0, 0, 0, 0, 0, damping),
        dof_stiffness=f(0, 0, 0, 0, 0, 0, stiffness), dof_springref=f(0, 0, 0, 0, 0, 0, springref),
        seg_body=i(0, 1), seg_pos=f([0, 0, 0], [0, 0, 0]), seg_quat=f([1, 0, 0, 0], [1, 0, 0, 0]),
        seg_invweight0=f([1.0 / base_mass, 1.0 / big], [1.0 / mass, 1.0 / inertia_yy]),
        act_type=i(0), act_trn=i(6), act_limited=i([limited, 0]), act_geom=i(-1), act_gain=f(kp if servo == "position" else kv), act_bias=f([-kp, -kv] if servo == "position" else [0.0, -kv]),
        act_forcerange=f([lo, hi]), act_ctrlrange=f([0.0, 0.0]), key_ctrl=f(0.0),
        key_qpos=f(0, 0, 100.0, 1, 0, 0, 0, q0), qpos0=f(0, 0, 100.0, 1, 0, 0, 0, q0),
        stat_meaninertia=f(mass),
    )
    if ctrlrange is not None:
        m.update(act_limited=i([limited, 1]), act_ctrlrange=f(list(ctrlrange)))
    if general is not None:
        # one of MuJoCo's general actuator shortcuts (intvelocity, damper, cylinder, muscle) on the hinge: the affine pass sees a
        # motor of gain 0 without a force limit, the row of act_general carries the rest (flygym_amd/compiler/model.py::_general_row)
        from flygym_amd.compiler.model import ACT_MOTOR, _general_row

        row = _general_row(dict(forcelimited=bool(limited), **general), 6)
        row[31] = abs(row[5]) / (inertia_yy + mass * float(np.dot(com, com)) + armature)      # acc0 = |M^-1 moment| at qpos0
        m.update(act_type=i(ACT_MOTOR), act_gain=f(0.0), act_bias=f([0.0, 0.0]), act_limited=i([0, 1 if ctrlrange is not None else 0]),
                 act_general=row.reshape(1, -1))
    return m


def welded_body(mass=1e-3, solref=(2e-4, 1.0), solimp=(0.98, 0.99, 1e-5, 0.5, 3.0), gravity=(0.0, 0.0, -9810.0), offset=(0.0, 0.0, 0.0),
                timestep=1e-4) -> CompiledModel:
    """One free body held by the tether weld (TetheredWorld's six bilateral rows, with the reference's weld parameters by
    default) to the pose (0, 0, 100) / identity, started ``offset`` away from it; no contact."""
    m = sphere_on_plane(mass=mass, radius=0.1, gravity=gravity, timestep=timestep, start_height=100.0)
    f, i = (lambda *a: np.asarray(a, dtype=np.float64)), (lambda *a: np.asarray(a, dtype=np.int32))
    start = np.array([0.0, 0.0, 100.0]) + np.asarray(offset, dtype=np.float64)
    big = 1e9 * mass
    m.update(weld_active=i(1), weld_params=f(0.0, 0.0, 100.0, 1, 0, 0, 0, *solref, *solimp, 1.0 / mass, 1.0 / big),
             key_qpos=f(*start, 1, 0, 0, 0), qpos0=f(*start, 1, 0, 0, 0))
    return m
