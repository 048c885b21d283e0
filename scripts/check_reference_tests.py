"""Run the reference's OWN unit tests against this package (API drop-in check).  Works only where the reference checkout
is mounted (this build container: /root/reference); nothing of the reference is copied into the repo — its test files
are collected from a temporary directory with `flygym` aliased to `flygym_amd` and `mujoco` / `dm_control` stubbed.

    python scripts/check_reference_tests.py [/root/reference]

Covered: tests/core/test_anatomy.py, test_physics.py, test_pose.py, test_utils.py, test_compose.py and
tests/examples/test_motion_snippet.py — the modules this build mirrors.  Expected to fail: whatever asserts MuJoCo / dm_control objects (isinstance(mj.MjModel), mjcf_root, video
and MJCF utilities) and test_fullsize_mesh_type (the snapshot's fullsize mesh folder has no c_head.stl: the reference
raises the same FileNotFoundError).  tests/core/test_simulation.py and tests/warp need a GPU and the reference side by
side, which never coexist; `tests/test_hip_parity.py` restates their invariants.
"""
import shutil, subprocess, sys, tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
ref = Path(sys.argv[1] if len(sys.argv) > 1 else "/root/reference")
if not (ref / "tests/core").is_dir():
    sys.exit(f"no reference checkout at {ref}")
CONFTEST = f'''
import sys, types, importlib, importlib.util
sys.path.insert(0, {str(ROOT)!r})
for _m in ("mujoco", "dm_control", "dm_control.mjcf"):
    sys.modules.setdefault(_m, types.ModuleType(_m))
sys.modules["dm_control"].mjcf = sys.modules["dm_control.mjcf"]
import flygym_amd
for k in ("", ".anatomy", ".compose", ".compose.fly", ".compose.world", ".compose.pose", ".compose.physics", ".utils",
          ".utils.math", ".utils.exceptions", ".utils.profiling", ".utils.pose_conversion", ".simulation"):
    sys.modules["flygym" + k] = importlib.import_module("flygym_amd" + k)
for k, v in (("flygym_demo", None), ("flygym_demo.spotlight_data", None), ("flygym_demo.spotlight_data.preprocessing", "flygym_amd.replay")):
    sys.modules[k] = importlib.import_module(v) if v else types.ModuleType(k)
sys.modules["flygym_demo.spotlight_data"].MotionSnippet = sys.modules["flygym_demo.spotlight_data.preprocessing"].MotionSnippet
spec = importlib.util.spec_from_file_location("ref_conftest", {str(ref / "tests/conftest.py")!r})
mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
globals().update({{k: v for k, v in vars(mod).items() if not k.startswith("__")}})
'''
with tempfile.TemporaryDirectory() as tmp:
    tmp = Path(tmp)
    (tmp / "conftest.py").write_text(CONFTEST)
    names = ["test_anatomy.py", "test_physics.py", "test_pose.py", "test_utils.py", "test_compose.py"]
    for n in names:
        shutil.copy(ref / "tests/core" / n, tmp / n)
    shutil.copy(ref / "tests/examples/test_motion_snippet.py", tmp / "test_motion_snippet.py")
    names.append("test_motion_snippet.py")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "--no-header", "-p", "no:cacheprovider", "-rf", *names], cwd=tmp)
    sys.exit(0 if r.returncode in (0, 1) else r.returncode)
