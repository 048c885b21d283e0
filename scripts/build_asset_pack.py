"""Build ``flygym_amd/assets/nmf_assets.npz`` from a flygym asset directory.

The engine needs the fly's *data* (rigging table, neutral poses, per-mesh rigid-body
constants and convex hulls, the Spotlight replay clip) at run time, but the GPU box
has no flygym checkout.  This script reads those data files where flygym keeps them

    <assets>/model/rigging.yaml                         (fly.py:32,549-550)
    <assets>/model/mujoco_globals.yaml                  (fly.py:33, utils/mjcf.py:31-43)
    <assets>/model/pose/neutral/<axis_order>.yaml       (pose.py:131-161)
    <assets>/model/meshes/{simplified_max2000faces,fullsize}/*.stl   (fly.py:507-543)
    <demo>/spotlight_data/assets/spotlight_behavior_clip.npz         (preprocessing.py:44-57)

and stores *derived numbers only* (no file is copied): per mesh the volume, COM,
inertia and convex hull after the reference's x1000 scale; the YAML scalars; the clip's
joint-angle array.  Re-run when the upstream assets change:

    python scripts/build_asset_pack.py [--assets DIR] [--clip FILE]
"""

import argparse
import json
import sys
from pathlib import Path

import numpy as np
import yaml

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))

from flygym_amd.anatomy import ALL_SEGMENT_NAMES  # noqa: E402
from flygym_amd.compiler.mesh import derive_mesh_data, load_binary_stl  # noqa: E402

SCALE = 1000.0  # fly.py:508-510: lengths are simulated in mm


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--assets", default="/root/reference/src/flygym/assets")
    ap.add_argument(
        "--clip",
        default="/root/reference/src/flygym_demo/spotlight_data/assets/spotlight_behavior_clip.npz",
    )
    ap.add_argument("--out", default=str(ROOT / "flygym_amd/assets/nmf_assets.npz"))
    args = ap.parse_args()
    assets = Path(args.assets)

    pack = {}
    rig = yaml.safe_load((assets / "model/rigging.yaml").read_text())
    pack["rigging_names"] = np.array(ALL_SEGMENT_NAMES)
    pack["rigging_pos"] = np.array([rig[n]["pos"] for n in ALL_SEGMENT_NAMES], dtype=np.float64)
    pack["rigging_quat"] = np.array([rig[n]["quat"] for n in ALL_SEGMENT_NAMES], dtype=np.float64)
    pack["rigging_mass"] = np.array([rig[n]["mass"] for n in ALL_SEGMENT_NAMES], dtype=np.float64)

    glob = yaml.safe_load((assets / "model/mujoco_globals.yaml").read_text())
    glob.pop("visual", None)
    pack["mujoco_globals_json"] = np.array(json.dumps(glob))

    poses = {}
    for f in sorted((assets / "model/pose/neutral").glob("*.yaml")):
        poses[f.stem] = yaml.safe_load(f.read_text())
    pack["neutral_pose_json"] = np.array(json.dumps(poses))
    # the same poses as YAML files next to the pack: `flygym_amd.assets_dir / "model/pose/neutral/<axis order>.yaml"`
    # is the path the reference's tutorials hand to KinematicPose(path=...)
    pose_dir = Path(args.out).parent / "model/pose/neutral"
    pose_dir.mkdir(parents=True, exist_ok=True)
    for stem, doc in poses.items():
        (pose_dir / f"{stem}.yaml").write_text(yaml.safe_dump(doc, sort_keys=True))

    for mesh_type in ("simplified_max2000faces", "fullsize"):
        d = assets / "model/meshes" / mesh_type
        names = []
        for f in sorted(d.glob("*.stl")):
            md = derive_mesh_data(load_binary_stl(f), scale=(SCALE, SCALE, SCALE))
            key = f"mesh/{mesh_type}/{f.stem}"
            names.append(f.stem)
            pack[key + "/props"] = np.concatenate(
                [[md.volume, md.hull_volume, md.n_vertices, md.n_faces], md.com, md.inertia.ravel()]
            )
            pack[key + "/hull_v"] = md.hull_vertices
            pack[key + "/hull_f"] = md.hull_faces
            print(f"{mesh_type:>24s} {f.stem:20s} V={md.volume:.4e} hullV={md.hull_volume:.4e} "
                  f"nv={md.n_vertices} nh={len(md.hull_vertices)}")
        pack[f"mesh/{mesh_type}/names"] = np.array(names)

    clip = np.load(args.clip, allow_pickle=True)
    pack["clip_joint_angles"] = clip["joint_angles"].astype(np.float32)
    pack["clip_legs"] = np.array([str(x) for x in clip["legs"].tolist()])
    pack["clip_dofs_per_leg"] = np.array([[str(y) for y in x] for x in clip["dofs_per_leg"].tolist()])
    pack["clip_fps"] = np.array(float(clip["data_fps"].item()))
    pack["clip_rawpred_egoxyz"] = clip["rawpred_egoxyz"].astype(np.float32)      # keypoint tracks of the recording
    pack["clip_fwdkin_egoxyz"] = clip["fwdkin_egoxyz"].astype(np.float32)
    pack["clip_keypoints"] = np.array([[str(y) for y in x] for x in clip["keypoints"].tolist()])
    pack["clip_experiment_trial"] = np.array(str(clip["experiment_trial"].item()))
    pack["clip_framerange"] = np.asarray(clip["framerange_in_raw_recording"], dtype=np.int64)

    out = Path(args.out)
    out.parent.mkdir(parents=True, exist_ok=True)
    np.savez_compressed(out, **pack)
    print("wrote", out, out.stat().st_size, "bytes")


if __name__ == "__main__":
    main()
