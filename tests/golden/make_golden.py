"""Generate golden vectors by IMPORTING the reference's mujoco-free Python modules.

Runs only in the build container (needs /root/reference); the outputs are data
fixtures committed next to this script.  Nothing here is imported by the product.

    python tests/golden/make_golden.py

Produces
  anatomy.json        orders / counts from reference ``anatomy.py``
  contact_params.json ``ContactParams()`` tuples from reference ``compose/physics.py``
  neutral_pose_ypr.json  neutral pose (rad, mirrored) from ``compose/pose.py`` loaders
  replay_42.npz       ``MotionSnippet().get_joint_angles(1e-4, actuated_order)``:
                      first 2000 rows (f32) + every 100th row + sha256 of the full f32 table
"""

import hashlib
import importlib.util
import json
import sys
import types
from pathlib import Path

import numpy as np

REF = Path("/root/reference/src")
OUT = Path(__file__).parent


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def import_reference():
    jt = types.ModuleType("jaxtyping")
    jt.Float = type("F", (), {"__getitem__": lambda s, k: object})()
    sys.modules["jaxtyping"] = jt
    fg = types.ModuleType("flygym")
    fg.__path__ = [str(REF / "flygym")]
    fg.assets_dir = REF / "flygym" / "assets"
    sys.modules["flygym"] = fg
    for pkg in ("utils", "compose"):
        p = types.ModuleType(f"flygym.{pkg}")
        p.__path__ = [str(REF / "flygym" / pkg)]
        sys.modules[f"flygym.{pkg}"] = p
    _load("flygym.utils.exceptions", REF / "flygym/utils/exceptions.py")
    _load("flygym.utils.math", REF / "flygym/utils/math.py")
    anatomy = _load("flygym.anatomy", REF / "flygym/anatomy.py")
    physics = _load("flygym.compose.physics", REF / "flygym/compose/physics.py")
    # pose.py needs py3.12-free syntax only; it imports yaml/numpy + flygym.anatomy
    pose = _load("flygym.compose.pose", REF / "flygym/compose/pose.py")
    demo = types.ModuleType("flygym_demo")
    demo.__path__ = [str(REF / "flygym_demo")]
    sys.modules["flygym_demo"] = demo
    sd = types.ModuleType("flygym_demo.spotlight_data")
    sd.__path__ = [str(REF / "flygym_demo/spotlight_data")]
    sys.modules["flygym_demo.spotlight_data"] = sd
    prep = _load(
        "flygym_demo.spotlight_data.preprocessing",
        REF / "flygym_demo/spotlight_data/preprocessing.py",
    )
    return anatomy, physics, pose, prep


def main():
    A, P, POSE, PREP = import_reference()

    # ---- anatomy orders -------------------------------------------------
    full = A.Skeleton(joint_preset=A.JointPreset.ALL_POSSIBLE, axis_order=A.AxisOrder.DONTCARE)
    body_order = ["c_thorax"]
    for d in full.iter_jointdofs("c_thorax"):
        if d.axis == A.RotationAxis.PITCH:
            body_order.append(d.child.name)
    out = {
        "all_segment_names": list(A.ALL_SEGMENT_NAMES),
        "all_connected_segment_pairs": [list(p) for p in A.ALL_CONNECTED_SEGMENT_PAIRS],
        "legs": list(A.LEGS),
        "bodysegs_order": body_order,
        "dof_orders": {},
        "actuated": {},
        "contact_presets": {},
        "axis_order_letters": {o.name: o.to_letters_xyz() for o in A.AxisOrder},
    }
    for preset in A.JointPreset:
        for order in ("YAW_PITCH_ROLL", "PITCH_ROLL_YAW", "ROLL_YAW_PITCH"):
            sk = A.Skeleton(joint_preset=preset, axis_order=A.AxisOrder[order])
            out["dof_orders"][f"{preset.value}/{order}"] = [d.name for d in sk.iter_jointdofs()]
    sk = A.Skeleton(joint_preset=A.JointPreset.LEGS_ONLY, axis_order=A.AxisOrder.YAW_PITCH_ROLL)
    for ap in A.ActuatedDOFPreset:
        out["actuated"][ap.value] = [d.name for d in sk.get_actuated_dofs_from_preset(ap)]
    for cp in A.ContactBodiesPreset:
        out["contact_presets"][cp.value] = [s.name for s in cp.to_body_segments_list()]
    (OUT / "anatomy.json").write_text(json.dumps(out, indent=1))

    # ---- contact params -------------------------------------------------
    cp = P.ContactParams()
    (OUT / "contact_params.json").write_text(
        json.dumps(
            {
                "friction": list(cp.get_friction_tuple()),
                "solref": list(cp.get_solref_tuple()),
                "solimp": list(cp.get_solimp_tuple()),
                "margin": cp.margin,
            },
            indent=1,
        )
    )

    # ---- neutral pose ---------------------------------------------------
    poses = {}
    for order in A.AxisOrder:
        if order.name in ("PRY", "PYR", "RPY", "RYP", "YPR", "YRP", "DONTCARE"):
            continue
        kp = POSE.KinematicPosePreset.NEUTRAL.get_pose_by_axis_order(order)
        poses[order.to_str()] = kp.joint_angles_lookup_rad
    (OUT / "neutral_pose.json").write_text(json.dumps(poses, indent=1, sort_keys=True))

    # ---- replay table ---------------------------------------------------
    actuated = sk.get_actuated_dofs_from_preset(A.ActuatedDOFPreset.LEGS_ACTIVE_ONLY)
    snippet = PREP.MotionSnippet(REF / "flygym_demo/spotlight_data/assets/spotlight_behavior_clip.npz")
    table = snippet.get_joint_angles(1e-4, actuated)
    t32 = np.ascontiguousarray(table.astype(np.float32))
    np.savez_compressed(
        OUT / "replay_42.npz",
        head=t32[:2000],
        every100=t32[::100],
        shape=np.array(table.shape),
        minmax=np.array([table.min(), table.max()]),
        sha256_f32=np.frombuffer(hashlib.sha256(t32.tobytes()).digest(), dtype=np.uint8),
        raw_first_frame=snippet.joint_angles[0].astype(np.float32),
    )
    print("golden fixtures written to", OUT)


if __name__ == "__main__":
    main()
