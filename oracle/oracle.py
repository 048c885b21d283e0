"""ctypes binding of the CPU oracle (``oracle/nmf_oracle.c``).  TEST INFRASTRUCTURE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg import
this module; nothing under ``flygym_amd/`` does.
"""

from __future__ import annotations

import ctypes
import subprocess
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent


def build(force: bool = False) -> None:
    libs = [HERE / "libnmf_oracle_f64.so", HERE / "libnmf_oracle_f32.so"]
    src = HERE / "nmf_oracle.c"
    if force or any((not l.exists()) or l.stat().st_mtime < src.stat().st_mtime for l in libs):
        subprocess.run(["make", "-C", str(HERE), "-s", "-B"], check=True, stderr=subprocess.DEVNULL)


_libs = {}


def _lib(precision: str):
    if precision not in _libs:
        path = HERE / f"libnmf_oracle_{precision}.so"
        if not path.exists():
            build()
        lib = ctypes.CDLL(str(path))
        sfx = "_" + precision
        for name, res, args in [
            ("nmfo_model_create", ctypes.c_void_p, [ctypes.c_char_p, ctypes.c_int64]),
            ("nmfo_model_dims", None, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int)]),
            ("nmfo_data_create", ctypes.c_void_p, [ctypes.c_void_p]),
            ("nmfo_forward", None, [ctypes.c_void_p, ctypes.c_void_p]),
            ("nmfo_step", None, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]),
            ("nmfo_reset", None, [ctypes.c_void_p, ctypes.c_void_p]),
            ("nmfo_step_replay", None, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                        ctypes.c_void_p, ctypes.c_int, ctypes.c_int]),
            ("nmfo_ptr", ctypes.c_void_p, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_int)]),
            ("nmfo_set_solver_mode", None, [ctypes.c_void_p, ctypes.c_int]),
            ("nmfo_set_noslip", None, [ctypes.c_void_p, ctypes.c_int]),
            ("nmfo_set_max_contacts", None, [ctypes.c_void_p, ctypes.c_int]),
            ("nmfo_ints", None, [ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]),
        ]:
            fn = getattr(lib, name + sfx)
            fn.restype, fn.argtypes = res, args
        _libs[precision] = lib
    return _libs[precision]


class Oracle:
    """One world stepped on the CPU.  ``precision`` is ``"f64"`` or ``"f32"``."""

    def __init__(self, model_blob: bytes, precision: str = "f64", cpu_flavour: bool = False):
        """``cpu_flavour``: run the model's ``noslip_iterations`` after the Newton solve, as the reference's CPU class does
        (``src/flygym/simulation.py:74-76`` with ``mujoco_globals.yaml:15``); the default is the batched class's behaviour,
        which strips the option (``src/flygym/warp/simulation.py:427-448``)."""
        self.precision = precision
        self.dtype = np.float64 if precision == "f64" else np.float32
        self._lib = _lib(precision)
        self._sfx = "_" + precision
        self._blob = bytes(model_blob)
        self._m = self._call("nmfo_model_create", self._blob, len(self._blob))
        if not self._m:
            raise ValueError("bad model blob")
        dims = (ctypes.c_int * 10)()
        self._call("nmfo_model_dims", self._m, dims)
        (self.nq, self.nv, self.nu, self.nb, self.nseg, self.ng, self.nsite, self.maxcon,
         self.nsensor, _) = list(dims)
        self._d = self._call("nmfo_data_create", self._m)
        self.cpu_flavour = bool(cpu_flavour)
        self._call("nmfo_set_noslip", self._d, int(self.cpu_flavour))
        self.reset()

    def set_max_contacts(self, n: int):
        """Contacts kept per step (``HIPSimulation(max_contacts=...)``; at most the engine's 48): later ones, in geom order, are dropped."""
        self._call("nmfo_set_max_contacts", self._d, int(n))

    def _call(self, name, *args):
        return getattr(self._lib, name + self._sfx)(*args)

    def clone_data(self) -> "Oracle":
        other = Oracle(self._blob, self.precision, self.cpu_flavour)
        for k in ("qpos", "qvel", "ctrl", "qacc_warmstart"):
            other.arr(k)[:] = self.arr(k)
        other.arr("time")[:] = self.arr("time")
        return other

    def arr(self, name: str) -> np.ndarray:
        """Writable view of an engine array (flat)."""
        n = ctypes.c_int(0)
        p = self._call("nmfo_ptr", self._m, self._d, name.encode(), ctypes.byref(n))
        if not p and n.value == 0:
            if name in ("con_dist", "con_pos", "con_frame", "efc_force", "efc_aref", "efc_D", "J", "site_xpos"):
                return np.zeros(0, dtype=self.dtype)
            raise KeyError(name)
        ctype = ctypes.c_double if self.precision == "f64" else ctypes.c_float
        return np.ctypeslib.as_array(ctypes.cast(p, ctypes.POINTER(ctype)), shape=(n.value,))

    def ints(self):
        out = (ctypes.c_int * 4)()
        geoms = (ctypes.c_int * self.maxcon)()
        self._call("nmfo_ints", self._m, self._d, out, geoms)
        ncon = out[0]
        return dict(ncon=ncon, nefc=out[1], overflow=out[2], solver_iter=out[3], con_geom=list(geoms)[:ncon])

    def set_solver_mode(self, mode: str):
        """``"shared"``: the stopping rules the HIP kernel also uses (rounding-floor exits, early line-search exit);
        ``"documented"``: MuJoCo-documented tolerance tests only, line search to its fixed point."""
        self._call("nmfo_set_solver_mode", self._d, {"shared": 0, "documented": 1}[mode])

    def reset(self):
        self._call("nmfo_reset", self._m, self._d)

    def forward(self):
        self._call("nmfo_forward", self._m, self._d)

    def step(self, n: int = 1):
        self._call("nmfo_step", self._m, self._d, int(n))

    def step_replay(self, table: np.ndarray, act_ids: np.ndarray, start: int, n: int):
        """table: (table_steps, n_act) float32; act_ids: (n_act,) int32."""
        table = np.ascontiguousarray(table, dtype=np.float32)
        act_ids = np.ascontiguousarray(act_ids, dtype=np.int32)
        self._call("nmfo_step_replay", self._m, self._d, table.ctypes.data, table.shape[0], table.shape[1],
                   act_ids.ctypes.data, int(start), int(n))

    # convenience ----------------------------------------------------------
    @property
    def qpos(self): return self.arr("qpos")
    @property
    def qvel(self): return self.arr("qvel")
    @property
    def ctrl(self): return self.arr("ctrl")
    @property
    def time(self): return float(self.arr("time")[0])
