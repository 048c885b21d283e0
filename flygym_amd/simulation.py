"""``HIPSimulation`` — the batched MI355X simulation, drop-in for the reference's
``flygym.warp.GPUSimulation`` (reference ``src/flygym/warp/simulation.py:28-453``; method
contracts from ``src/flygym/simulation.py:16-480``).

Same method names, argument orders, batch-leading shapes and fly-ordered columns.  State queries
return ``torch`` tensors on the GPU (the reference returns ``wp.array``); control inputs accept
numpy arrays or torch tensors.  All physics runs in ``libnmf_hip.so`` (hand-written HIP for
gfx950); there is no CPU path here.
"""

from __future__ import annotations

import ctypes
import warnings

import numpy as np

from . import _native
from .compose.fly import ActuatorType
from .compose.world import BaseWorld

__all__ = ["HIPSimulation"]


class HIPSimulation:
    """Runs ``n_worlds`` copies of one world in lock-step on one MI355X.

    Args:
        world: a configured :class:`~flygym_amd.compose.BaseWorld` with one fly.
        n_worlds: number of parallel worlds on this GPU.
        max_contacts: contacts kept per world and step, as ``GPUSimulation``'s ``max_contacts`` sizes MJWarp's
            contact arrays (``warp/simulation.py:50-56``) — up to the engine's 48 (lane = contact; 192 pyramid rows in
            registers / LDS): ``contact_capacity = min(max_contacts, 48)``.  Contacts beyond the capacity, in geom
            order, are dropped and the step counts as overflowed (:meth:`get_solver_stats` column 2;
            :meth:`overflow_steps`); MJWarp drops them too but only prints.  ``contact_bound`` is the most contacts
            the model's contact set can make at once: a model with ``contact_bound <= contact_capacity`` cannot
            overflow; ``strict_contacts=True`` refuses any other model up front.
        max_constraints: accepted for signature compatibility (four pyramid rows per kept contact + the tether's six).
        strict_contacts: raise at construction if the model can make more contacts than the engine keeps.
        device: CUDA/HIP device index (one process per GPU for multi-GPU runs).
        _options: development switches of ``nmf_batch_create_ex`` (``include/nmf.h``; keywords of
            :meth:`flygym_amd._native.BatchOptions.make`: ``solver`` "primal" / "nohist" / "nofallback", ``sched`` "plain",
            ``order``, ``max_chunks``, ...).  What a batch runs is reported by :meth:`batch_info`.
    """

    def __new__(cls, world=None, *args, **kwargs):
        # a world with several flies (reference compose/world.py:95-149): one batch per fly behind the same surface
        if cls is HIPSimulation and world is not None and len(getattr(world, "fly_lookup", {})) > 1:
            return object.__new__(MultiFlyHIPSimulation)
        return object.__new__(cls)

    def __init__(self, world: BaseWorld, n_worlds: int, max_constraints: int = 500,
                 max_contacts: int = 500, device: int | None = None, strict_contacts: bool = False,
                 _cpu_flavour: bool = False, _options: dict | None = None) -> None:
        import torch

        if len(world.fly_lookup) == 0:
            raise ValueError("The world must contain at least one fly.")
        if int(max_contacts) < 1:
            raise ValueError(f"max_contacts must be at least 1, got {max_contacts}")
        if not _cpu_flavour:      # (flygym_amd.Simulation keeps the CPU class's noslip pass, on every skeleton and world)
            self._strip_unsupported_options(world)
        self.world = world
        self.n_worlds = int(n_worlds)
        self.max_constraints = max_constraints
        self.max_contacts = max_contacts
        self.renderer = None
        if not torch.cuda.is_available():
            raise _native.NativeError("HIPSimulation needs a visible MI355X (torch.cuda.is_available() is False)")
        self.device_index = torch.cuda.current_device() if device is None else int(device)
        if not 0 <= self.device_index < torch.cuda.device_count():
            raise _native.NativeError(f"no GPU with index {self.device_index} ({torch.cuda.device_count()} visible)")
        self.device = torch.device("cuda", self.device_index)
        self._torch = torch
        self._lib = _native.lib()

        self.model = world.compile_model()
        blob = self.model.to_blob()
        self._model_h = self._lib.nmf_model_create(blob, len(blob))
        if not self._model_h:
            raise _native.NativeError(self._lib.nmf_last_error().decode())
        opts = _native.BatchOptions.make(**(_options or {}))
        with torch.cuda.device(self.device):
            self._batch_h = self._lib.nmf_batch_create_ex(self._model_h, self.n_worlds, self.device_index, ctypes.byref(opts))
        if not self._batch_h:
            raise _native.NativeError(self._lib.nmf_last_error().decode())
        self.contact_bound = _native.check_count(self._lib.nmf_model_contact_bound(self._model_h))
        self.contact_capacity = _native.check_count(self._lib.nmf_batch_set_contact_capacity(self._batch_h, int(max_contacts)))
        if strict_contacts and self.contact_bound > self.contact_capacity:
            raise ValueError(f"this model's contact set can make {self.contact_bound} contacts in one step; the engine keeps "
                             f"{self.contact_capacity} per world (max_contacts={max_contacts}, engine limit 48)")
        self._views = {}
        self._build_index_maps()
        # the MuJoCo attributes reference code reads most often (the reference's GPUSimulation keeps a CPU mj_model /
        # mj_data of world 0 next to the device arrays); `mj_data` fields are float64 copies of world 0 off the GPU
        from types import SimpleNamespace

        m = self.model
        self.mj_model = SimpleNamespace(opt=SimpleNamespace(timestep=float(m["opt_timestep"][0])), nq=m.nq, nv=m.nv, nu=m.nu,
                                        nbody=m.nbody, njnt=m.njnt, nsite=m.nsite, compiled=m)
        self.mj_data = _DataView(self)
        self._curr_step = 0
        self._frames_rendered = 0
        self._profile_events = []
        self._physics_time_folded_ns = 0
        self._total_render_time_ns = 0

    # ---- lifecycle -----------------------------------------------------------------
    def __del__(self):
        try:
            if getattr(self, "_batch_h", None):
                self._lib.nmf_batch_destroy(self._batch_h)
                self._batch_h = None
            if getattr(self, "_model_h", None):
                self._lib.nmf_model_destroy(self._model_h)
                self._model_h = None
        except Exception:
            pass

    @staticmethod
    def _strip_unsupported_options(world: BaseWorld) -> bool:
        """The batched reference path runs without the noslip post-pass
        (``warp/simulation.py:427-448``); so does this engine."""
        if world.noslip_iterations > 0:
            warnings.warn(
                "The batched engine does not run noslip iterations. Changing "
                f"option/noslip_iterations from {world.noslip_iterations} to 0."
            )
            world.noslip_iterations = 0
            world._compiled = None
            return True
        return False

    def _stream(self):
        return ctypes.c_void_p(self._torch.cuda.current_stream(self.device).cuda_stream)

    def _ids(self, arr):
        return self._torch.as_tensor(np.asarray(arr, dtype=np.int32), device=self.device)

    def _build_index_maps(self):
        m = self.model
        self._ids_by_fly = {}
        for fly_name, fly in self.world.fly_lookup.items():
            nd = len(fly.get_jointdofs_order())
            maps = dict(
                qpos=self._ids(np.arange(7, 7 + nd)),
                qvel=self._ids(np.arange(6, 6 + nd)),
                bodies=self._ids(np.arange(m.nseg)),
                sites=self._ids(np.arange(m.nsite)),
                adhesion=self._ids([i for i, a in enumerate(fly.actuators) if a["kind"] == "adhesion"]),
                actuators={},
            )
            for ty in ActuatorType:
                ids = [i for i, a in enumerate(fly.actuators) if a["kind"] == ty.value and ty != ActuatorType.ADHESION]
                if ids:
                    maps["actuators"][ty] = self._ids(ids)
            self._ids_by_fly[fly_name] = maps

    # ---- raw views -------------------------------------------------------------------
    def field(self, name: str):
        """Zero-copy torch view ``(n_worlds, width)`` of an engine array (aliases device state)."""
        if name not in self._views:
            width = ctypes.c_int32(0)
            ptr = self._lib.nmf_field_ptr(self._batch_h, _native.FIELDS[name], ctypes.byref(width))
            if not ptr:
                raise _native.NativeError(self._lib.nmf_last_error().decode())
            view = _tensor_from_ptr(self._torch, ptr, (self.n_worlds, max(width.value, 0)), self.device)
            if name in _native.INT_FIELDS:       # integer counters behind the float-typed field pointer (include/nmf.h)
                view = view.view(self._torch.int32)
            self._views[name] = view
        return self._views[name]

    def _gather(self, field: str, ids, group: int = 1):
        n = int(ids.numel())
        shape = (self.n_worlds, n) if group == 1 else (self.n_worlds, n, group)
        dst = self._torch.empty(shape, dtype=self._torch.float32, device=self.device)
        if n:
            _native.check(self._lib.nmf_gather(self._batch_h, _native.FIELDS[field], ids.data_ptr(), n, group,
                                               dst.data_ptr(), self._stream()))
        return dst

    def _to_device(self, x, n_cols: int, what: str):
        """(n_worlds, n_cols) float32 on the device.  As on the reference's GPU class — whose scatter kernel reads the
        leading ``n_cols`` columns of whatever it is given (warp/simulation.py:236-258; its own tests pass one column per
        joint dof for 42 actuators, tests/warp/test_simulation.py:254-268) — a wider array is accepted and its leading
        columns are used; a narrower one, or a wrong number of worlds, is an error here (there: an out-of-bounds read)."""
        t = self._torch
        if not isinstance(x, t.Tensor):
            x = t.as_tensor(np.asarray(x, dtype=np.float32))
        if x.ndim != 2 or x.shape[0] != self.n_worlds or x.shape[1] < n_cols:
            raise ValueError(f"Expected {what} of shape ({self.n_worlds}, {n_cols}), but got {tuple(x.shape)}")
        if x.shape[1] > n_cols:
            x = x[:, :n_cols]
        return x.to(device=self.device, dtype=t.float32).contiguous()

    # ---- reference surface ---------------------------------------------------------------
    def reset(self) -> None:
        _native.check(self._lib.nmf_reset(self._batch_h, self._stream()))
        if self.renderer is not None:
            self.renderer.reset()
        self._curr_step = 0
        self._frames_rendered = 0
        self._total_physics_time_ns = 0
        self._total_render_time_ns = 0

    def reset_worlds(self, mask) -> None:
        """Reset only the worlds where ``mask`` (``(n_worlds,)`` bool, numpy or torch) is set: keyframe pose, zero
        velocity, clock at 0.  The other worlds keep their state.  Stream-ordered device work (no host sync)."""
        t = self._torch
        m = t.as_tensor(mask, device=self.device)
        if tuple(m.shape) != (self.n_worlds,):
            raise ValueError(f"Expected a reset mask of shape ({self.n_worlds},), but got {tuple(m.shape)}")
        m = (m != 0).to(t.uint8).contiguous()
        _native.check(self._lib.nmf_reset_worlds(self._batch_h, m.data_ptr(), self._stream()))

    def step(self, n_steps: int = 1, record_every: int | None = None, n_act: int = 42):
        """Advance all worlds by one timestep (``n_steps`` > 1 fuses several into one launch).

        ``record_every=k``: the launch also records the observation block of every k-th step — what the reference's loops read
        after every ``step()`` (``get_joint_angles`` / ``get_joint_velocities`` / ``get_actuator_forces`` /
        ``get_ground_contact_info``, reference ``simulation.py:142-243``) — and returns it as a float32 tensor
        ``(n_steps // k, n_worlds, 2 nj + n_act + 96)`` in the layout of :meth:`pack_observations`; row ``i`` is bit for bit what
        ``pack_observations`` gives after step ``(i + 1) k`` (``nmf_step_record``)."""
        if record_every is None:
            _native.check(self._lib.nmf_step(self._batch_h, int(n_steps), self._stream()))
            return None
        return self._record(None, None, 0, n_steps, record_every, n_act)

    def _record(self, table, act_ids, start, n_steps, record_every, n_act):
        t = self._torch
        k = int(record_every)
        if k < 1 or int(n_steps) < k or int(n_steps) % k:
            raise ValueError(f"record_every must be in 1..n_steps and divide n_steps, got {record_every} for {n_steps} steps")
        width = 2 * (self.model.nv - 6) + int(n_act) + 96
        ring = t.empty((int(n_steps) // k, self.n_worlds, width), dtype=t.float32, device=self.device)
        return self.record_into(ring, table, act_ids, start, n_steps, k, n_act)

    def record_into(self, ring, table, act_ids, start: int, n_steps: int, record_every: int, n_act: int = 42):
        """:meth:`step` / :meth:`step_replay` with ``record_every``, writing into a caller-owned ring (float32, contiguous,
        ``(n_steps // record_every, n_worlds, >= 2 nj + n_act + 96)`` on this device) — no allocation per tick."""
        t = self._torch
        nj = self.model.nv - 6
        width = 2 * nj + int(n_act) + 96
        k = int(record_every)
        if k < 1 or int(n_steps) < k or int(n_steps) % k:
            raise ValueError(f"record_every must be in 1..n_steps and divide n_steps, got {record_every} for {n_steps} steps")
        if ring.dtype != t.float32 or ring.device != self.device or ring.ndim != 3 or not ring.is_contiguous() \
                or ring.shape[0] < int(n_steps) // k or ring.shape[1] != self.n_worlds or ring.shape[2] < width:
            raise ValueError(f"the observation ring must be a contiguous float32 ({int(n_steps) // k}+, {self.n_worlds}, {width}+) tensor on {self.device}")
        tab = (table.data_ptr(), int(table.shape[1]), int(table.shape[2]), act_ids.data_ptr()) if table is not None else (None, 0, 0, None)
        _native.check(self._lib.nmf_step_record(self._batch_h, tab[0], tab[1], tab[2], tab[3], int(start), int(n_steps), k, nj, int(n_act),
                                                ring.data_ptr(), int(ring.shape[2]), self._stream()))
        return ring

    def step_replay(self, table, act_ids, start: int, n_steps: int, record_every: int | None = None, n_act: int = 42):
        """Device-resident replay loop: before step ``s`` load ``ctrl[:, act_ids] = table[:, start+s]``.
        ``record_every``: as in :meth:`step` (returns the observation ring).

        ``table``: float32 ``(n_worlds, table_steps, n_act)`` and ``act_ids``: int32 ``(n_act,)`` engine control ids
        (:meth:`replay_ids`), both contiguous on this simulation's device — the kernel reads them through raw pointers,
        so anything else is refused here."""
        t = self._torch
        if not (isinstance(table, t.Tensor) and isinstance(act_ids, t.Tensor)):
            raise ValueError("step_replay takes torch tensors on the simulation's device")
        if table.dtype != t.float32 or act_ids.dtype != t.int32:
            raise ValueError(f"step_replay needs a float32 table and int32 ids, got {table.dtype} / {act_ids.dtype}")
        if table.device != self.device or act_ids.device != self.device:
            raise ValueError(f"step_replay needs tensors on {self.device}, got {table.device} / {act_ids.device}")
        if table.ndim != 3 or table.shape[0] != self.n_worlds or act_ids.ndim != 1 or act_ids.numel() != table.shape[2]:
            raise ValueError(f"Expected a table of shape ({self.n_worlds}, table_steps, n_act) and n_act ids, "
                             f"but got {tuple(table.shape)} and {tuple(act_ids.shape)}")
        if not (table.is_contiguous() and act_ids.is_contiguous()):
            raise ValueError("step_replay needs contiguous tensors")
        if record_every is not None:
            return self._record(table, act_ids, start, n_steps, record_every, n_act)
        _native.check(self._lib.nmf_step_replay(
            self._batch_h, table.data_ptr(), int(table.shape[1]), int(table.shape[2]), act_ids.data_ptr(),
            int(start), int(n_steps), self._stream()))
        return None

    def replay_ids(self, fly_name: str, with_adhesion: bool = False):
        """Engine control ids (device int32) of the columns of a replay / CPG target table: the fly's position actuators
        in ``get_actuated_jointdofs_order`` order, then (optionally) its six adhesion actuators in ``LEGS`` order."""
        ids = self._ids_by_fly[fly_name]["actuators"][ActuatorType.POSITION]
        if with_adhesion:
            ids = self._torch.cat([ids, self._ids_by_fly[fly_name]["adhesion"]])
        return ids

    def step_with_profile(self) -> None:
        """One step bracketed by device events recorded on the launch stream (the reference brackets a synchronous
        ``mj_step`` with ``perf_counter_ns``, ``simulation.py:78-84``; a launch here is asynchronous, so host clocks
        would time the enqueue).  No host synchronisation: the events are read when the report is asked for."""
        t = self._torch
        e0, e1 = t.cuda.Event(enable_timing=True), t.cuda.Event(enable_timing=True)
        stream = t.cuda.current_stream(self.device)
        e0.record(stream)
        self.step()
        e1.record(stream)
        self._profile_events.append((e0, e1))
        if len(self._profile_events) >= 1024:
            self._fold_profile_events()
        self._curr_step += 1

    def _fold_profile_events(self) -> None:
        if self._profile_events:
            self._profile_events[-1][1].synchronize()
            self._physics_time_folded_ns += int(sum(a.elapsed_time(b) for a, b in self._profile_events) * 1e6)
            self._profile_events.clear()

    @property
    def _total_physics_time_ns(self) -> int:
        """GPU time of the steps taken with :meth:`step_with_profile` (synchronises on the last of them)."""
        self._fold_profile_events()
        return self._physics_time_folded_ns

    @_total_physics_time_ns.setter
    def _total_physics_time_ns(self, value: int) -> None:
        self._profile_events.clear()
        self._physics_time_folded_ns = int(value)

    def pack_observations(self, out, n_act: int = 42):
        """Write the observation block ``[joint angles | joint velocities | forces of the first n_act actuators | 96
        contact-sensor floats]`` of every world into the rows of ``out`` (float32, ``(>= n_worlds, >= width)``, unit
        column stride) in one launch (``nmf_pack_observations``): the input of the multi-GPU all-gather."""
        t = self._torch
        nj = self.model.nv - 6
        width = 2 * nj + n_act + 96
        if out.dtype != t.float32 or out.device != self.device or out.ndim != 2 or out.shape[0] < self.n_worlds \
                or out.shape[1] < width or out.stride(1) != 1:
            raise ValueError(f"pack_observations needs a float32 ({self.n_worlds}+, {width}+) tensor on {self.device}")
        _native.check(self._lib.nmf_pack_observations(self._batch_h, nj, int(n_act), out.data_ptr(), int(out.stride(0)), self._stream()))
        return out

    def shader_clock_hz(self, reset: bool = False) -> float:
        """Shader clock (Hz) the stepping launches since the last ``reset=True`` call ran at (``nmf_shader_clock``;
        synchronises).  0.0 if there were none."""
        hz = ctypes.c_double(0.0)
        _native.check(self._lib.nmf_shader_clock(self._batch_h, ctypes.byref(hz), int(reset)))
        return float(hz.value)

    def warmup(self, duration_s: float = 0.05) -> None:
        n = int(duration_s / self.timestep)
        if n > 0:
            self.step(n)

    def get_joint_angles(self, fly_name: str):
        return self._gather("qpos", self._ids_by_fly[fly_name]["qpos"])

    def get_joint_velocities(self, fly_name: str):
        return self._gather("qvel", self._ids_by_fly[fly_name]["qvel"])

    def get_body_positions(self, fly_name: str):
        return self._gather("seg_xpos", self._ids_by_fly[fly_name]["bodies"], 3)

    def get_body_rotations(self, fly_name: str):
        return self._gather("seg_xquat", self._ids_by_fly[fly_name]["bodies"], 4)

    def get_site_positions(self, fly_name: str):
        return self._gather("site_xpos", self._ids_by_fly[fly_name]["sites"], 3)

    def get_actuator_forces(self, fly_name: str, actuator_type):
        ids = self._ids_by_fly[fly_name]["actuators"][ActuatorType(actuator_type)]
        return self._gather("actuator_force", ids)

    def get_ground_contact_info(self, fly_name: str):
        """(active, force, torque, pos, normal, tangent) with a leading batch axis: shapes
        ``(n_worlds, 6)`` and ``(n_worlds, 6, 3)`` (reference ``simulation.py:210-243``)."""
        if self.world.legpos_to_groundcontactsensors_by_fly is None:
            raise ValueError("this world has no ground contact sensors")
        sd = self.field("sensordata").reshape(self.n_worlds, 6, 16).clone()
        return sd[:, :, 0], sd[:, :, 1:4], sd[:, :, 4:7], sd[:, :, 7:10], sd[:, :, 10:13], sd[:, :, 13:16]

    def get_solver_stats(self):
        """``(n_worlds, 8)`` of the last step: contacts, Newton iterations, contact-overflow flag, constraint rows, solve-report
        bits, most pivots of an elimination, KKT residual of the last elimination's target, 0 (``NMF_STATS``)."""
        return self.field("stats").clone()

    SOLVER_EXIT_NAMES = ("contact_space", "kkt_exact", "tie_rule", "stalled_line_search", "cost_tests", "iteration_limit",
                         "primal_loop", "fallback_resolves", "big_eliminations", "noslip_skipped", "no_contact")

    def get_solver_exits(self) -> dict:
        """How the constraint solves of all steps since the last reset ended, summed over the worlds (``NMF_STATS_SUM`` columns
        4..14; synchronises): ``steps`` and one count per kind."""
        tot = self.field("stats_sum").to(self._torch.int64).sum(dim=0).cpu().tolist()
        out = {"steps": int(tot[0])}
        out.update({k: int(tot[4 + i]) for i, k in enumerate(self.SOLVER_EXIT_NAMES)})
        return out

    def batch_info(self) -> dict:
        """What this batch runs (``nmf_batch_info``): kernel family, contact-space flavour and the contacts it takes, flies per
        CU, chunk plan, world-order policy, solver option bits, kernel LDS / VGPRs."""
        out = (ctypes.c_int32 * 16)()
        _native.check(self._lib.nmf_batch_info(self._batch_h, out))
        return dict(zip(_native.INFO_KEYS, [int(v) for v in out]))

    def overflow_steps(self) -> int:
        """Steps since the last reset, summed over the worlds, in which a world made more contacts than
        ``contact_capacity`` and the surplus (highest geom indices) was dropped.  Synchronises."""
        return int(self.field("stats_sum")[:, 3].sum().item())

    def set_actuator_inputs(self, fly_name: str, actuator_type, inputs) -> None:
        ids = self._ids_by_fly[fly_name]["actuators"][ActuatorType(actuator_type)]
        n = int(ids.numel())
        if len(inputs.shape) == 2 and inputs.shape[1] < n:
            raise ValueError(
                f"Expected {n} inputs for actuator type '{ActuatorType(actuator_type).name}', but got {inputs.shape[1]}"
            )
        src = self._to_device(inputs, n, "actuator inputs")
        _native.check(self._lib.nmf_scatter(self._batch_h, _native.FIELDS["ctrl"], ids.data_ptr(), n,
                                            src.data_ptr(), self._stream()))

    def set_leg_adhesion_states(self, fly_name: str, leg_to_adhesion_state) -> None:
        ids = self._ids_by_fly[fly_name]["adhesion"]
        n = int(ids.numel())
        if len(leg_to_adhesion_state.shape) == 2 and leg_to_adhesion_state.shape[1] < n:
            raise ValueError(
                f"Unexpected number of adhesion states: expected {n}, got {leg_to_adhesion_state.shape[1]}"
            )
        src = self._to_device(leg_to_adhesion_state, n, "adhesion states")
        _native.check(self._lib.nmf_scatter(self._batch_h, _native.FIELDS["ctrl"], ids.data_ptr(), n,
                                            src.data_ptr(), self._stream()))

    @property
    def time(self) -> float:
        """Current simulation time in seconds (from world 0; synchronises)."""
        return float(self.field("time")[0, 0].item())

    @property
    def timestep(self) -> float:
        return float(self.model["opt_timestep"][0])

    # ---- rendering hand-off (out of scope for the engine) ------------------
    def set_renderer(self, *args, **kwargs):
        raise NotImplementedError(
            "rendering is outside the stepping engine: read poses with get_body_positions/"
            "get_body_rotations and hand them to a renderer of your choice"
        )

    def render_as_needed(self):
        if self.renderer is None:
            return {}
        return self.renderer.render_as_needed(self)

    def print_performance_report(self) -> None:
        """Report of the steps taken with :meth:`step_with_profile` (reference ``warp/simulation.py:344-367``)."""
        from .utils.profiling import print_perf_report_parallel

        print_perf_report_parallel(self._total_physics_time_ns, self._total_render_time_ns, self._curr_step,
                                   self._frames_rendered, self.timestep, self.n_worlds, 0)


class MultiFlyHIPSimulation(HIPSimulation):
    """``HIPSimulation(world, n_worlds)`` of a world that holds several flies (reference ``compose/world.py:95-149``;
    ``GPUSimulation`` / ``Simulation`` address every query and input by ``fly_name``, ``simulation.py:142-243``).

    The reference gives fly geoms ``contype = conaffinity = 0`` and adds fly-ground contact pairs only
    (``compose/fly.py:609-610``, ``compose/world.py:300-309``): the flies of a world never touch each other, and MuJoCo's
    constraint problem of such a world is block-diagonal — one block per fly, the blocks' optima independent (what they share in
    MuJoCo is the stopping test of one joint Newton loop: tolerance-level).  So each fly is stepped as its own batch of
    ``n_worlds`` worlds — ``sims[fly_name]``, an ordinary :class:`HIPSimulation` of ``world.single_fly_view(fly_name)`` — and this
    class routes the per-fly calls, and fans ``step`` / ``reset`` / ``warmup`` out to all of them (launches of different flies are
    independent kernels on the caller's stream).  Engine-level accessors that belong to one batch (``field``, ``step_replay``,
    ``pack_observations``, ...) are reached through :meth:`for_fly`."""

    def __init__(self, world: BaseWorld, n_worlds: int, max_constraints: int = 500,
                 max_contacts: int = 500, device: int | None = None, strict_contacts: bool = False,
                 _cpu_flavour: bool = False, _options: dict | None = None) -> None:
        if not _cpu_flavour:
            self._strip_unsupported_options(world)          # once, on the world itself: the views inherit it
        self.world = world
        self.n_worlds = int(n_worlds)
        self.max_constraints, self.max_contacts = max_constraints, max_contacts
        self.renderer = None
        self.sims = {name: HIPSimulation(world.single_fly_view(name), n_worlds, max_constraints, max_contacts, device, strict_contacts,
                                         _cpu_flavour, _options) for name in world.fly_lookup}
        first = next(iter(self.sims.values()))
        self.device, self.device_index, self._torch = first.device, first.device_index, first._torch
        self._frames_rendered = 0
        self._total_render_time_ns = 0

    def __del__(self):
        pass

    def for_fly(self, fly_name: str) -> HIPSimulation:
        """The batch that steps ``fly_name`` (its ``field`` views, ``step_replay``, ``batch_info``, ...)."""
        return self.sims[fly_name]

    def _each(self):
        return self.sims.values()

    # ---- fanned out --------------------------------------------------------------------
    def reset(self) -> None:
        for s in self._each(): s.reset()

    def reset_worlds(self, mask) -> None:
        for s in self._each(): s.reset_worlds(mask)

    def step(self, n_steps: int = 1, record_every: int | None = None, n_act: int = 42):
        out = {name: s.step(n_steps, record_every, n_act) for name, s in self.sims.items()}
        return None if record_every is None else out

    def step_with_profile(self) -> None:
        for s in self._each(): s.step_with_profile()

    def warmup(self, duration_s: float = 0.05) -> None:
        for s in self._each(): s.warmup(duration_s)

    @property
    def _curr_step(self): return next(iter(self._each()))._curr_step

    @property
    def _total_physics_time_ns(self): return sum(s._total_physics_time_ns for s in self._each())

    @property
    def time(self) -> float: return next(iter(self._each())).time

    @property
    def timestep(self) -> float: return next(iter(self._each())).timestep

    def overflow_steps(self) -> int: return sum(s.overflow_steps() for s in self._each())

    def get_solver_stats(self): return {name: s.get_solver_stats() for name, s in self.sims.items()}

    def get_solver_exits(self) -> dict: return {name: s.get_solver_exits() for name, s in self.sims.items()}

    def batch_info(self) -> dict: return {name: s.batch_info() for name, s in self.sims.items()}

    def print_performance_report(self) -> None:
        from .utils.profiling import print_perf_report_parallel

        print_perf_report_parallel(self._total_physics_time_ns, self._total_render_time_ns, self._curr_step,
                                   self._frames_rendered, self.timestep, self.n_worlds, 0)

    # ---- routed by fly name ------------------------------------------------------------
    def get_joint_angles(self, fly_name: str): return self.sims[fly_name].get_joint_angles(fly_name)
    def get_joint_velocities(self, fly_name: str): return self.sims[fly_name].get_joint_velocities(fly_name)
    def get_body_positions(self, fly_name: str): return self.sims[fly_name].get_body_positions(fly_name)
    def get_body_rotations(self, fly_name: str): return self.sims[fly_name].get_body_rotations(fly_name)
    def get_site_positions(self, fly_name: str): return self.sims[fly_name].get_site_positions(fly_name)
    def get_actuator_forces(self, fly_name: str, actuator_type): return self.sims[fly_name].get_actuator_forces(fly_name, actuator_type)
    def get_ground_contact_info(self, fly_name: str): return self.sims[fly_name].get_ground_contact_info(fly_name)
    def set_actuator_inputs(self, fly_name: str, actuator_type, inputs) -> None: self.sims[fly_name].set_actuator_inputs(fly_name, actuator_type, inputs)
    def set_leg_adhesion_states(self, fly_name: str, leg_to_adhesion_state) -> None: self.sims[fly_name].set_leg_adhesion_states(fly_name, leg_to_adhesion_state)
    def replay_ids(self, fly_name: str, with_adhesion: bool = False): return self.sims[fly_name].replay_ids(fly_name, with_adhesion)

    # ---- one batch's business ------------------------------------------------------------
    def _one_batch_only(self, what):
        raise AttributeError(f"{what} belongs to one fly's batch: use sim.for_fly(fly_name).{what} ({', '.join(self.sims)})")

    def field(self, name: str): self._one_batch_only("field")
    def step_replay(self, *a, **k): self._one_batch_only("step_replay")
    def record_into(self, *a, **k): self._one_batch_only("record_into")
    def pack_observations(self, *a, **k): self._one_batch_only("pack_observations")
    def shader_clock_hz(self, *a, **k): self._one_batch_only("shader_clock_hz")

    @property
    def model(self): self._one_batch_only("model")

    @property
    def mj_model(self): self._one_batch_only("mj_model")

    @property
    def mj_data(self): self._one_batch_only("mj_data")


def _tensor_from_ptr(torch, ptr: int, shape, device):
    """Wrap a raw device pointer as a torch tensor without copying."""
    n = int(np.prod(shape))

    class _Holder:
        pass

    h = _Holder()
    h.__cuda_array_interface__ = {
        "shape": (max(n, 1),), "typestr": "<f4", "data": (int(ptr), False), "version": 3, "strides": None,
    }
    t = torch.as_tensor(h, device=device)
    return t[:n].reshape(shape)


class _DataView:
    """Read-only stand-in for ``mj_data`` on :class:`Simulation`: every attribute is a fresh float64 copy of world 0."""

    _FIELDS = dict(qpos="qpos", qvel="qvel", ctrl="ctrl", qacc="qacc", qacc_warmstart="qacc_warmstart",
                   actuator_force="actuator_force", sensordata="sensordata")

    def __init__(self, batch):
        self._batch = batch

    def __getattr__(self, name):
        if name in self._FIELDS:
            return self._batch.field(self._FIELDS[name])[0].cpu().numpy().astype(np.float64)
        if name == "site_xpos":
            return self._batch.field("site_xpos")[0].cpu().numpy().astype(np.float64).reshape(-1, 3)
        if name == "time":
            return self._batch.time
        raise AttributeError(f"mj_data.{name} is not available from the HIP engine; see HIPSimulation.field()")


class Simulation:
    """Single-world simulation with the reference's CPU ``Simulation`` surface (``src/flygym/simulation.py:16-480``):
    unbatched numpy in, unbatched float64 numpy out — the physics still runs in the HIP engine (one wavefront).

    For user loops written against ``flygym.Simulation``; throughput work belongs on :class:`HIPSimulation`
    (a single world is launch-latency bound: ≈ 0.1 ms per ``step()``; use ``step(n)`` to fuse steps).  Differences:
    no ``mj_model`` / ``mj_data`` (the engine arrays are reachable through ``batch.field(name)``), rendering handed off.
    The CPU class's noslip post-pass (``option/noslip_iterations = 5``, ``mujoco_globals.yaml:15``) runs here too, on every
    skeleton and world: inside the contact-space solve where a step takes it (leg-chain skeletons and ALL_BIOLOGICAL with up
    to 16 / 13 contacts on the legs), else after the primal Newton loop with ``A = J M^-1 J^T`` built by one articulated-body
    solve per constraint row (``csrc/nmf_step.hip::noslip_primal`` — one world: cost is no object).  A step with contacts
    that went without the pass would be counted (``get_solver_exits()["noslip_skipped"]``); none does.
    """

    def __init__(self, world: BaseWorld, device: int | None = None) -> None:
        self.batch = HIPSimulation(world, 1, device=device, _cpu_flavour=True)
        self.world = world
        self.renderer = None
        # (a world with several flies has one compiled model per fly: batch.for_fly(name).mj_model)
        self.mj_model, self.mj_data = getattr(self.batch, "mj_model", None), getattr(self.batch, "mj_data", None)

    # profiling counters of the reference class (simulation.py:52-56), kept by the batch object
    @property
    def _curr_step(self): return self.batch._curr_step
    @property
    def _frames_rendered(self): return self.batch._frames_rendered
    @property
    def _total_physics_time_ns(self): return self.batch._total_physics_time_ns
    @property
    def _total_render_time_ns(self): return self.batch._total_render_time_ns

    # -- stepping
    def reset(self) -> None:
        self.batch.reset()

    def step(self, n_steps: int = 1) -> None:
        self.batch.step(n_steps)

    def step_with_profile(self) -> None:
        self.batch.step_with_profile()

    def warmup(self, duration_s: float = 0.05) -> None:
        self.batch.warmup(duration_s)

    # -- state (fly order, as the reference)
    @staticmethod
    def _np(t):
        return t[0].cpu().numpy().astype(np.float64)

    def get_joint_angles(self, fly_name: str) -> np.ndarray:
        return self._np(self.batch.get_joint_angles(fly_name))

    def get_joint_velocities(self, fly_name: str) -> np.ndarray:
        return self._np(self.batch.get_joint_velocities(fly_name))

    def get_body_positions(self, fly_name: str) -> np.ndarray:
        return self._np(self.batch.get_body_positions(fly_name))

    def get_body_rotations(self, fly_name: str) -> np.ndarray:
        return self._np(self.batch.get_body_rotations(fly_name))

    def get_site_positions(self, fly_name: str) -> np.ndarray:
        return self._np(self.batch.get_site_positions(fly_name))

    def get_actuator_forces(self, fly_name: str, actuator_type) -> np.ndarray:
        return self._np(self.batch.get_actuator_forces(fly_name, actuator_type))

    def get_ground_contact_info(self, fly_name: str):
        """(active (6,), force (6, 3), torque, pos, normal, tangent) as in ``simulation.py:210-243``."""
        return tuple(self._np(x) for x in self.batch.get_ground_contact_info(fly_name))

    # -- controls
    def set_actuator_inputs(self, fly_name: str, actuator_type, inputs) -> None:
        ids = self.batch._ids_by_fly[fly_name]["actuators"][ActuatorType(actuator_type)]
        if len(inputs) != int(ids.numel()):
            raise ValueError(
                f"Expected {int(ids.numel())} inputs for actuator type '{ActuatorType(actuator_type).name}', but got {len(inputs)}"
            )
        self.batch.set_actuator_inputs(fly_name, actuator_type, np.asarray(inputs, dtype=np.float32)[None, :])

    def set_leg_adhesion_states(self, fly_name: str, leg_to_adhesion_state) -> None:
        ids = self.batch._ids_by_fly[fly_name]["adhesion"]
        if len(leg_to_adhesion_state) != int(ids.numel()):
            raise ValueError(
                f"Unexpected number of adhesion states: expected {int(ids.numel())}, got {len(leg_to_adhesion_state)}"
            )
        self.batch.set_leg_adhesion_states(fly_name, np.asarray(leg_to_adhesion_state, dtype=np.float32)[None, :])

    # -- rendering is handed off, as on HIPSimulation
    def set_renderer(self, *args, **kwargs):
        return self.batch.set_renderer(*args, **kwargs)

    def render_as_needed(self) -> bool:
        return self.batch.render_as_needed()

    def render_as_needed_with_profile(self) -> bool:
        return self.batch.render_as_needed()

    def print_performance_report(self) -> None:
        self.batch.print_performance_report()

    @property
    def time(self) -> float:
        return self.batch.time

    @property
    def timestep(self) -> float:
        return self.batch.timestep
