"""One-step accuracy of the default constraint solve on states that do NOT depend on it (run through gpurun; NMF_HIP_LIB selects the
library): 4096 walking flies are rolled out on the primal Newton loop (round 3's solver: its code is the same in every library
variant, so two variants see bit-identical states), then at several times the states are pushed into a default-solver batch, stepped
once, and `qacc` of sampled worlds is compared with the float64 oracle stepped from the same state.
usage: python scripts/onestep_error.py [--worlds 4096] [--samples 48] [--checkpoints 6] [--terrain flat|blocks|mixed] [--preset legs_only]"""
import argparse, json, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "oracle"))
import numpy as np, torch
from flygym_amd import HIPSimulation, make_model
from flygym_amd.controllers import TripodCPG
import oracle as orc

ap = argparse.ArgumentParser()
ap.add_argument("--worlds", type=int, default=4096); ap.add_argument("--samples", type=int, default=48)
ap.add_argument("--checkpoints", type=int, default=6); ap.add_argument("--terrain", default="flat"); ap.add_argument("--preset", default="legs_only")
a = ap.parse_args()
n = a.worlds


def world():
    fly, w, _ = make_model(joints_preset=a.preset)
    if a.terrain != "flat":
        import flygym_amd.compose as C
        from flygym_amd.utils.math import Rotation3D
        w = {"gapped": C.GappedTerrainWorld, "blocks": C.BlocksTerrainWorld, "mixed": C.MixedTerrainWorld}[a.terrain]()
        w.add_fly(fly, (0, 0, 0.8), Rotation3D("quat", (1, 0, 0, 0)))
    return fly, w


fly, w1 = world()
gen = HIPSimulation(w1, n_worlds=n, device=0, _options=dict(solver="primal"))
fly2, w2 = world()
sim = HIPSimulation(w2, n_worlds=n, device=0)
for s in (gen, sim): s.set_leg_adhesion_states(fly.name, np.ones((n, 6), dtype=np.float32))
gen.warmup()
table = TripodCPG(fly.get_actuated_jointdofs_order("position"), gen.timestep).targets(n, 2500, device=gen.device)
ids = gen.replay_ids(fly.name)
gen.step_replay(table, ids, 0, 400)
keys = ("qpos", "qvel", "ctrl", "qacc_warmstart")
blob = sim.model.to_blob()
devs, its = [], []
cur = 400
rng = np.random.default_rng(0)
for cp in range(a.checkpoints):
    gen.step_replay(table, ids, cur, 61); cur += 61
    for k in keys: sim.field(k)[:] = gen.field(k)
    sim.step_replay(table, ids, cur, 2)           # (the first step builds the active-set history, the second is judged)
    torch.cuda.synchronize()
    # the state before the judged step: re-create it by stepping the generator's state once on `sim`'s own arithmetic is not possible
    # without the history; instead judge step 2 from sim's state after step 1
    for k in keys: sim.field(k)[:] = gen.field(k)
    sim.step_replay(table, ids, cur, 1)
    torch.cuda.synchronize()
    state = {k: sim.field(k).cpu().numpy().astype(np.float64) for k in keys}
    sim.step_replay(table, ids, cur + 1, 1)
    torch.cuda.synchronize()
    qacc, stats, geom = sim.field("qacc").cpu().numpy().astype(np.float64), sim.field("stats").cpu().numpy(), sim.field("contact_geom").cpu().numpy()
    its.append(float(stats[:, 1].mean()))
    tab = table.cpu().numpy()
    for wd in rng.choice(n, size=a.samples, replace=False):
        o = orc.Oracle(blob, "f64")
        for k in keys: o.arr(k)[:] = state[k][wd]
        o.step_replay(tab[wd], ids.cpu().numpy(), cur + 1, 1)
        nc = int(stats[wd, 0])
        if nc != o.ints()["ncon"] or geom[wd, :nc].astype(int).tolist() != o.ints()["con_geom"]: continue
        ref = o.arr("qacc")
        devs.append(float(np.abs(qacc[wd] - ref).max() / max(np.abs(ref).max(), 1e4)))
devs = np.array(devs)
print(json.dumps(dict(lib=str(__import__("flygym_amd")._native.LIB_PATH.name), compared=len(devs), median=float(np.median(devs)), p90=float(np.quantile(devs, 0.9)),
                      p99=float(np.quantile(devs, 0.99)), worst=float(devs.max()), iters=float(np.mean(its)), exits=sim.get_solver_exits())))
