"""The reference's OWN test files, run unchanged against this package through an import alias (scripts/run_reference_tests.py).

Only where the reference is present (this container; never on the GPU box): the files are read from /root/reference at run
time and nothing of them is kept in this repository.  What must hold: every test of the reference's anatomy, pose, contact
parameter and motion-snippet suites passes on ``flygym_amd``; of its compose and utils suites everything passes except the
tests listed here, each of which needs a MuJoCo / dm_control object this package deliberately does not build (MJCF elements,
``compile()`` to an ``MjModel``, the full-size meshes of the offline asset tooling, video files).
"""

import json
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]

NEEDS_MUJOCO = {
    "core/test_compose.py": {
        "TestFlatGroundWorld::test_compile_returns_mujoco_model", "TestTetheredWorld::test_compile_returns_mujoco_model",
        "TestFlyCompile::test_compile_produces_mujoco_model",                       # isinstance(..., mujoco.MjModel)
        "TestFlatGroundWorld::test_custom_name", "TestFlyColorize::test_colorize_adds_materials",
        "TestFlyAddTrackingCamera::test_camera_full_identifier_after_world_attachment",     # dm_control MJCF elements
        "TestFlyConstructionOptions::test_fullsize_mesh_type",                      # full-size meshes: not in the asset pack
    },
    "core/test_utils.py": {
        "TestSetMujocoGlobals::test_applies_yaml_settings", "TestSetMujocoGlobals::test_missing_yaml_raises",
        "TestSetParamsRecursive::test_non_dict_child_raises", "TestSetParamsRecursive::test_sets_attribute_on_root",
        "TestSetParamsRecursive::test_sets_nested_attribute", "TestSetParamsRecursive::test_unknown_key_is_silently_ignored",   # dm_control
        "TestWriteVideoFromFrames::test_creates_file", "TestWriteVideoFromFrames::test_creates_parent_dirs",
        "TestWriteVideoFromFrames::test_multiple_of_16_written_unchanged", "TestWriteVideoFromFrames::test_non_multiple_of_16_is_resized",   # video files
    },
}
MIN_PASSED = {"core/test_anatomy.py": 64, "core/test_pose.py": 30, "core/test_physics.py": 23, "core/test_compose.py": 54,
              "core/test_utils.py": 50, "examples/test_motion_snippet.py": 12}


def test_the_references_own_tests_pass_on_this_package():
    if not Path("/root/reference/tests").exists():
        pytest.skip("the reference is not present on this machine")
    r = subprocess.run([sys.executable, str(ROOT / "scripts" / "run_reference_tests.py")], capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-2000:]
    res = json.loads(r.stdout)
    for rel, want in MIN_PASSED.items():
        got = res[rel]
        assert set(got["failed_ids"]) <= NEEDS_MUJOCO.get(rel, set()), f"{rel}: unexpected failures {sorted(set(got['failed_ids']) - NEEDS_MUJOCO.get(rel, set()))}"
        assert got["passed"] >= want, f"{rel}: {got['passed']} passed, expected at least {want}"
