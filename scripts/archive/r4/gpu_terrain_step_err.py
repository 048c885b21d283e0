"""One-step error of the engine against the float64 oracle on a terrain world, per world, with contact / wall counts
(diagnostic, run through gpurun)."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "oracle"))
import numpy as np, torch
import flygym_amd.compose as C
from flygym_amd import HIPSimulation
from flygym_amd import anatomy as A
from flygym_amd.controllers import TripodCPG
from flygym_amd.utils.math import Rotation3D
import oracle as orc
cls = getattr(C, sys.argv[1] if len(sys.argv) > 1 else "BlocksTerrainWorld")
fly = C.Fly(name="t")
sk = A.Skeleton(axis_order=A.AxisOrder.YAW_PITCH_ROLL, joint_preset=A.JointPreset.LEGS_ONLY)
fly.add_joints(sk, neutral_pose=C.KinematicPosePreset.NEUTRAL)
fly.add_actuators(sk.get_actuated_dofs_from_preset("legs_active_only"), C.ActuatorType.POSITION, kp=50.0, neutral_input=C.KinematicPosePreset.NEUTRAL)
fly.add_leg_adhesion()
world = cls()
world.add_fly(fly, (0, 0, 0.8), Rotation3D("quat", (1, 0, 0, 0)))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 256
sim = HIPSimulation(world, n_worlds=n, device=0)
cpg = TripodCPG(fly.get_actuated_jointdofs_order("position"), 1e-4)
table = cpg.targets(n, 2500, device=sim.device)
ids = sim.replay_ids(fly.name)
sim.set_leg_adhesion_states(fly.name, np.ones((n, 6), dtype=np.float32))
sim.warmup(); sim.step_replay(table, ids, 0, 600)
seen_walls = 0; seen = 0
blob = sim.model.to_blob()
bad = 0
for rep in range(4):
    sim.step_replay(table, ids, 600 + 50 * rep, 49)
    torch.cuda.synchronize()
    state = {k: sim.field(k).cpu().numpy().astype(np.float64) for k in ("qpos", "qvel", "ctrl", "qacc_warmstart")}
    sim.step_replay(table, ids, 600 + 50 * rep + 49, 1)
    torch.cuda.synchronize()
    qacc = sim.field("qacc").cpu().numpy(); st = sim.field("stats").cpu().numpy(); geom = sim.field("contact_geom").cpu().numpy()
    for w in range(0, n, max(2, n // 128)):
        o = orc.Oracle(blob, "f64")
        o.qpos[:] = state["qpos"][w]; o.qvel[:] = state["qvel"][w]; o.arr("qacc_warmstart")[:] = state["qacc_warmstart"][w]; o.ctrl[:] = state["ctrl"][w]
        o.step_replay(table[w].cpu().numpy(), ids.cpu().numpy(), 600 + 50 * rep + 49, 1)
        nc = int(st[w, 0])
        if o.ints()["ncon"] != nc: continue
        a = o.arr("qacc"); err = np.abs(qacc[w] - a).max() / np.abs(a).max()
        fr = o.arr("con_frame").reshape(-1, 9)
        nwall = int((np.abs(fr[:, 2]) < 0.5).sum())
        g = geom[w, :nc].astype(int)
        dup = int(max(np.bincount(g)))
        seen += 1; seen_walls += nwall > 0
        if err > 2e-3:
            bad += 1
            if bad < 25: print(f"rep {rep} world {w}: ncon {nc} walls {nwall} max contacts per geom {dup} iters {int(st[w,1])}/{o.ints()['solver_iter']} err {err:.2e}")
print("bad", bad, "of", seen, "sampled; with wall contacts", seen_walls)
