"""Debug: where do HIP eye frames differ from the numpy specification (run through gpurun)."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "oracle")); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np, torch
import sensors_oracle as so
import flygym_amd.compose as C
from flygym_amd import HIPSimulation, make_model
from flygym_amd.utils.math import Rotation3D
from flygym_amd.vision import EyeRenderer, Scene
from test_sensors import _world_capsules

for world_cls in sys.argv[1:] or ["BlocksTerrainWorld"]:
    fly, world, _ = make_model()
    world = getattr(C, world_cls)()
    world.add_fly(fly, (0.4, 0.1, 0.8), Rotation3D("quat", (1, 0, 0, 0)))
    n = 2
    sim = HIPSimulation(world, n_worlds=n, device=0)
    sim.field("qvel")[:, :6] = torch.as_tensor(np.random.default_rng(7).normal(0, 15, (n, 6)), dtype=torch.float32, device=sim.device)
    sim.set_leg_adhesion_states(fly.name, np.ones((n, 6), dtype=np.float32))
    sim.step(250)
    scene = Scene(spheres=[(6.0, 4.0, 1.5, 1.0)], sphere_rgb=[(0.9, 0.2, 0.1)])
    eyes = EyeRenderer(sim, fly.name, scene)
    fr1 = eyes.render_frames().cpu().numpy()
    fr2 = eyes.render_frames().cpu().numpy()
    print(world_cls, "render twice identical:", np.array_equal(fr1, fr2), "qpos", sim.field("qpos")[0, :7].cpu().numpy())
    names = [s.name for s in fly.get_bodysegs_order()]
    xpos = sim.field("seg_xpos").cpu().numpy().reshape(n, 69, 3).astype(np.float64)
    xquat = sim.field("seg_xquat").cpu().numpy().reshape(n, 69, 4).astype(np.float64)
    tp = sim.model["terrain_params"]
    terrain = (int(sim.model["terrain_type"][0]), tuple(float(v) for v in tp[:4]), float(tp[4]))
    pal = {"sky": scene.sky_rgb, "gA": scene.ground_rgb[0], "gB": scene.ground_rgb[1], "wall": scene.wall_rgb, "body": scene.body_rgb, "sphere": scene.sphere_rgb[0], "black": (0, 0, 0)}
    def name_of(img):
        out = np.full(img.shape[:2], "?", dtype=object)
        for k, c in pal.items():
            out[(img == np.array(c, dtype=np.uint8)).all(axis=-1)] = k
        return out
    for w in range(n):
        caps = _world_capsules(eyes, xpos[w], xquat[w])
        for e, (seg, pos, quat) in enumerate(eyes.cameras):
            Rs = so.quat_to_mat(xquat[w, names.index(seg)])
            cam = xpos[w, names.index(seg)] + Rs @ pos
            want = so.render_eye_frames(cam, Rs @ so.quat_to_mat(quat), 512, 450, 157.0, 4.0, 0.0, scene.sky_rgb, scene.ground_rgb,
                                        scene.spheres, scene.sphere_rgb, terrain=terrain, wall_rgb=scene.wall_rgb, capsules=caps, body_rgb=scene.body_rgb)
            d = (fr1[w, e] != want).any(axis=-1)
            a, b = name_of(fr1[w, e])[d], name_of(want)[d]
            import collections
            print(f" world {w} eye {e}: diff {d.mean():.4f} cam {np.round(cam, 3)}", collections.Counter(zip(a.tolist(), b.tolist())).most_common(6))
            rows, cols = np.where(d)
            if len(rows): print("   rows", rows.min(), rows.max(), "cols", cols.min(), cols.max())
