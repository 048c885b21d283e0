"""Run the REFERENCE's own test files against this package (dev-time check, this container only).

The reference's tests (`/root/reference/tests`) exercise its public API — anatomy, poses, contact parameters, the compose
layer, the motion-snippet data path.  This script runs those files UNCHANGED against `flygym_amd` through an import alias
(`flygym` -> `flygym_amd`, `flygym_demo.spotlight_data.preprocessing` -> `flygym_amd.replay`): they are read from
`/root/reference` at run time and copied to a temporary directory only (one line is patched there: `import mujoco`, which
this image lacks) — nothing of them is stored in this repository.  Prints one JSON object {file: {passed, failed, failed_ids}}.
Tests that need MuJoCo / dm_control objects (MJCF elements, `compile()` to an MjModel, video writing) fail by design: this
package records the model choices instead of building MJCF; tests/test_reference_suite.py pins the list.
"""
import json, re, shutil, subprocess, sys, tempfile
from pathlib import Path

REF = Path("/root/reference/tests")
ROOT = Path(__file__).resolve().parents[1]
FILES = ["core/test_anatomy.py", "core/test_pose.py", "core/test_physics.py", "core/test_compose.py", "core/test_utils.py",
         "examples/test_motion_snippet.py"]
BOOT = r'''
import sys, importlib, types
sys.path.insert(0, {root!r})
import flygym_amd
sys.modules["flygym"] = flygym_amd
for sub in ["anatomy", "compose", "compose.fly", "compose.world", "compose.pose", "compose.physics", "utils", "utils.math",
            "simulation", "warp", "utils.profiling", "utils.exceptions"]:
    try:
        sys.modules["flygym." + sub] = importlib.import_module("flygym_amd." + sub)
    except ImportError:
        pass
sys.modules["flygym_demo"] = types.ModuleType("flygym_demo")
sys.modules["flygym_demo.spotlight_data"] = types.ModuleType("flygym_demo.spotlight_data")
sys.modules["flygym_demo.spotlight_data.preprocessing"] = importlib.import_module("flygym_amd.replay")
import pytest
sys.exit(pytest.main(["-q", "--no-header", "-p", "no:cacheprovider", "-rf"] + sys.argv[1:]))
'''


def main():
    if not REF.exists():
        print(json.dumps({"skipped": "no /root/reference here"}))
        return 0
    out = {}
    with tempfile.TemporaryDirectory(prefix="nmf_reftests_") as tmp:
        tmp = Path(tmp)
        shutil.copytree(REF, tmp / "tests", ignore=shutil.ignore_patterns("__pycache__"))
        for f in (tmp / "tests").rglob("*.py"):
            src = f.read_text()
            if re.search(r"^import mujoco as mj$", src, flags=re.M):
                f.write_text(re.sub(r"^import mujoco as mj$", "mj = None", src, flags=re.M))
        (tmp / "run.py").write_text(BOOT.format(root=str(ROOT)))
        for rel in FILES:
            r = subprocess.run([sys.executable, "run.py", f"tests/{rel}"], cwd=tmp, capture_output=True, text=True, timeout=900)
            tail = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else ""
            passed = int(m.group(1)) if (m := re.search(r"(\d+) passed", tail)) else 0
            failed = int(m.group(1)) if (m := re.search(r"(\d+) failed", tail)) else 0
            ids = sorted(re.findall(r"^FAILED tests/\S+?::(\S+)", r.stdout, flags=re.M))
            out[rel] = {"passed": passed, "failed": failed, "failed_ids": ids}
    print(json.dumps(out, indent=1))
    return 0


if __name__ == "__main__":
    sys.exit(main())
