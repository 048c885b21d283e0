"""ctypes binding of the C ABI in ``include/nmf.h`` (``libnmf_hip.so``).

The product path has no CPU fallback: if the HIP library is missing or no MI355X is visible,
loading / batch creation raises.
"""

from __future__ import annotations

import ctypes
import os
import subprocess
from pathlib import Path

PKG = Path(__file__).resolve().parent
# NMF_HIP_LIB: load another build of the same ABI (kernel A/B experiments: scripts/build_variant.sh); the default is the
# in-tree library that build() compiles
LIB_PATH = Path(os.environ["NMF_HIP_LIB"]) if os.environ.get("NMF_HIP_LIB") else PKG / "libnmf_hip.so"
MATH_FLAGS = ["-fno-hip-fp32-correctly-rounded-divide-sqrt", "-freciprocal-math", "-fno-signed-zeros", "-fassociative-math", "-fno-trapping-math", "-fno-math-errno", "-fapprox-func"]
CSRC = PKG / "csrc"
INCLUDE = PKG.parent / "include"

FIELDS = dict(
    qpos=0, qvel=1, ctrl=2, qacc_warmstart=3, seg_xpos=4, seg_xquat=5, site_xpos=6,
    actuator_force=7, sensordata=8, time=9, stats=10, qacc=11, cost=12, stats_sum=13, contact_geom=14, act=15,
)

INT_FIELDS = frozenset({"stats_sum"})   # fields whose 32-bit words are unsigned integer counters, not floats

_lib = None


class NativeError(RuntimeError):
    pass


class BatchOptions(ctypes.Structure):
    """``nmf_batch_options`` of include/nmf.h (0 = the library's default)."""

    _fields_ = [("struct_size", ctypes.c_int32), ("solver", ctypes.c_int32), ("sched", ctypes.c_int32), ("order", ctypes.c_int32),
                ("max_chunks", ctypes.c_int32), ("min_chunk_steps", ctypes.c_int32), ("order_every", ctypes.c_int32),
                ("rest_slow", ctypes.c_int32), ("chunk_div", ctypes.c_float), ("flies_per_cu", ctypes.c_int32)]

    SOLVER = {"": 0, "default": 0, "primal": 1, "nohist": 2, "nofallback": 4}
    SCHED = {"": 0, "chunks": 0, "plain": 1}
    ORDER = {"": 0, "auto": 0, "inorder": 1, "costliest": 2, "none": 3, "policy": 4}

    @classmethod
    def make(cls, solver="", sched="", order="", max_chunks=0, min_chunk_steps=0, order_every=0, rest_slow=False, chunk_div=0.0,
             flies_per_cu=0):
        return cls(ctypes.sizeof(cls), cls.SOLVER[solver], cls.SCHED[sched], cls.ORDER[order], int(max_chunks), int(min_chunk_steps),
                   int(order_every), int(bool(rest_slow)), float(chunk_div), int(flies_per_cu))


INFO_KEYS = ("kernel_family", "terrain_kernel", "tether_kernel", "contact_space_flavour", "contact_space_max_contacts", "flies_per_cu",
             "resident_workgroups", "chunked", "max_chunks", "chunk_div_x1000", "order_policy", "solver_option_bits", "noslip_iterations",
             "contact_capacity", "kernel_lds_bytes", "kernel_vgprs")


def build(force: bool = False, verbose: bool = False) -> Path:
    """Compile the HIP engine for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    srcs = [CSRC / "nmf_capi.hip", CSRC / "nmf_step.hip", CSRC / "nmf_sensors.hip", CSRC / "nmf_eyes.hip", CSRC / "nmf_replay.hip", CSRC / "nmf_device.h", CSRC / "nmf_tree.h", CSRC / "nmf_dual.h",
            INCLUDE / "nmf.h", Path(__file__)]   # this file holds the compiler flags
    if os.environ.get("NMF_HIP_LIB"):
        return LIB_PATH                       # an externally built variant: nothing to compile here
    def fresh():
        return LIB_PATH.exists() and all(LIB_PATH.stat().st_mtime >= s.stat().st_mtime for s in srcs)

    if not force and fresh():
        return LIB_PATH
    # one builder at a time: the ranks of a multi-GPU launch import this together, and a library that is being written must
    # never be the one another rank maps (build into a temporary file, rename when done)
    import fcntl

    lock = open(LIB_PATH.with_suffix(".lock"), "w")
    fcntl.flock(lock, fcntl.LOCK_EX)
    try:
        if not force and fresh():
            return LIB_PATH                   # another process built it while this one waited
        return _compile(verbose)
    finally:
        fcntl.flock(lock, fcntl.LOCK_UN)
        lock.close()


def _compile(verbose: bool) -> Path:
    # -fno-slp-vectorize: the SLP vectoriser packs the 6-vector arithmetic into v_pk_* pairs and pays for it in v_mov
    # shuffles and register pressure (12 spilled VGPRs); scalar code is 9 % faster on the step kernel.  The iterative
    # ILP scheduler interleaves the independent chains of the unrolled sweeps better than the default (+5 %).
    # Arithmetic: divisions and square roots to 1-2.5 ulp (v_rcp / v_sqrt without the correctly-rounded fix-up sequences),
    # reassociation and no signed zeros (+3.7 % together); NaN / Inf semantics are kept (no -ffinite-math-only), and the
    # only transcendental of the step, the joint-angle sincos, is the kernel's own polynomial (nmf_device.h).
    # No atomic optimizer: the kernels' atomics are one-lane ticket / counter operations; the optimizer rewrites a returning
    # one into a wave-wide form that waits for the value at once (the chunk scheduler requests its ticket ahead of use).
    tmp = LIB_PATH.with_name(LIB_PATH.name + f".{os.getpid()}.tmp")
    cmd = [
        "hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-mllvm", "-amdgpu-sched-strategy=iterative-ilp",
        *MATH_FLAGS, "-mllvm", "-amdgpu-atomic-optimizer-strategy=None", "-fPIC", "-shared",
        f"-I{INCLUDE}", f"-I{CSRC}", str(CSRC / "nmf_capi.hip"), "-o", str(tmp),
    ]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        tmp.unlink(missing_ok=True)
        raise NativeError("hipcc failed:\n" + res.stderr[-4000:])
    if verbose:
        print(res.stderr)
    os.replace(tmp, LIB_PATH)
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise NativeError(
                f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(the MI355X engine has no CPU fallback)"
            )
        L = ctypes.CDLL(str(LIB_PATH))
        vp, ci, cf = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
        sig = {
            "nmf_last_error": (ctypes.c_char_p, []),
            "nmf_model_create": (vp, [ctypes.c_char_p, ctypes.c_size_t]),
            "nmf_model_destroy": (None, [vp]),
            "nmf_model_dims": (ci, [vp, ctypes.POINTER(ctypes.c_int32)]),
            "nmf_batch_create": (vp, [vp, ci, ci]),
            "nmf_batch_create_ex": (vp, [vp, ci, ci, vp]),
            "nmf_batch_info": (ci, [vp, ctypes.POINTER(ctypes.c_int32)]),
            "nmf_step_record": (ci, [vp, vp, ci, ci, vp, ci, ci, ci, ci, ci, vp, ci, vp]),
            "nmf_batch_destroy": (None, [vp]),
            "nmf_batch_n_worlds": (ci, [vp]),
            "nmf_model_contact_bound": (ci, [vp]),
            "nmf_batch_set_contact_capacity": (ci, [vp, ci]),
            "nmf_reset": (ci, [vp, vp]),
            "nmf_reset_worlds": (ci, [vp, vp, vp]),
            "nmf_step": (ci, [vp, ci, vp]),
            "nmf_step_replay": (ci, [vp, vp, ci, ci, vp, ci, ci, vp]),
            "nmf_field_ptr": (vp, [vp, ci, ctypes.POINTER(ctypes.c_int32)]),
            "nmf_gather": (ci, [vp, ci, vp, ci, ci, vp, vp]),
            "nmf_scatter": (ci, [vp, ci, vp, ci, vp, vp]),
            "nmf_pack_observations": (ci, [vp, ci, ci, vp, ci, vp]),
            "nmf_step_count": (ctypes.c_int64, [vp]),
            "nmf_shader_clock": (ci, [vp, ctypes.POINTER(ctypes.c_double), ci]),
            "nmf_time_launches": (ctypes.c_double, [vp, vp, ci, ci, vp, ci, ci, vp]),
            "nmf_retina_plan_bytes": (ctypes.c_size_t, [ci]),
            "nmf_retina_plan": (ci, [vp, ci, vp, vp]),
            "nmf_retina_resample": (ci, [vp, vp, vp, vp, vp, ci, ci, ci, vp, vp]),
            "nmf_eye_params_size": (ctypes.c_size_t, []),
            "nmf_eye_render": (ci, [vp, vp, vp, vp, vp, vp, vp, vp, vp, ci, vp, vp, vp]),
            "nmf_eye_plan_create": (vp, [vp, vp, vp, vp, ci, ci, cf, ci, ci]),
            "nmf_eye_plan_destroy": (None, [vp]),
            "nmf_eye_render_planned": (ci, [vp, vp, vp, vp, vp, vp, vp, vp, vp]),
            "nmf_odor_intensity": (ci, [vp, vp, vp, ci, vp, vp, ci, ci, vp, vp]),
            "nmf_replay_resample": (ci, [vp, ci, ci, ctypes.c_double, ctypes.c_double, vp, ci, ci, vp, vp]),
        }
        for name, (res, args) in sig.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


def check(rc: int) -> None:
    if rc != 0:
        raise NativeError(lib().nmf_last_error().decode() or "libnmf_hip call failed")


def check_count(rc: int) -> int:
    """For entry points that return a count (>= 0) or a negative error code."""
    if rc < 0:
        raise NativeError(lib().nmf_last_error().decode() or "libnmf_hip call failed")
    return int(rc)


def exported_symbols() -> list[str]:
    """Names declared in include/nmf.h (used by the CPU-side ABI test)."""
    import re

    text = (INCLUDE / "nmf.h").read_text()
    return sorted(set(re.findall(r"\b(nmf_[a-z_]+)\s*\(", text)))
