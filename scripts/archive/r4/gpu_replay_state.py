"""Step ONE saved state (gpu_blowup_probe.py / gpu_config5_probe.py format) once on every solver variant and print what comes
out (diagnostic, run through gpurun): python scripts/r4/gpu_replay_state.py <blocks|mixed> state.npz [...]"""
import os, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import numpy as np, torch
import flygym_amd.compose as C
from flygym_amd import HIPSimulation, make_model
from flygym_amd.utils.math import Rotation3D
kind = sys.argv[1]
def mk(solver):
    fly = make_model()[0]
    world = {"blocks": C.BlocksTerrainWorld, "mixed": C.MixedTerrainWorld, "flat": C.FlatGroundWorld}[kind]()
    world.add_fly(fly, (0, 0, 0.8), Rotation3D("quat", (1, 0, 0, 0)))
    if solver: os.environ["NMF_SOLVER"] = solver
    else: os.environ.pop("NMF_SOLVER", None)
    sim = HIPSimulation(world, n_worlds=4, device=0)
    os.environ.pop("NMF_SOLVER", None)
    return fly, sim
for f in sys.argv[2:]:
    d = np.load(f)
    for solver in ("", "nohist", "primal"):
        fly, sim = mk(solver)
        ids = sim.replay_ids(fly.name, with_adhesion=d["rows"].shape[1] > 42)
        rows = torch.as_tensor(d["rows"], device=sim.device)[None].repeat(4, 1, 1).contiguous()
        for k in ("qpos", "qvel", "ctrl", "qacc_warmstart"):
            sim.field(k)[:] = torch.as_tensor(d[k], device=sim.device)[None, :]
        sim.step_replay(rows, ids, int(d["cur"]), 1)
        torch.cuda.synchronize()
        st = sim.field("stats")[0].cpu().numpy(); qa = sim.field("qacc")[0].cpu().numpy()
        print(f"{Path(f).name} solver {solver or 'default':8s}: contacts {st[0]:.0f} iterations {st[1]:.0f} max|qacc| {np.abs(qa).max():.4g}  (saved kernel qacc max {np.abs(d['qacc_kernel']).max():.4g})")
