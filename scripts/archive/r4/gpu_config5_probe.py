"""Config 5 (mixed terrain + 20x gait adhesion): the engine's own states after 300 steps, one more step on each solver variant
against the float64 oracle, every world (diagnostic, run through gpurun).  Saves the worst state to gpurun_out/."""
import os, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "oracle"))
import numpy as np, torch
import oracle as orc
import flygym_amd.compose as C
from flygym_amd import HIPSimulation, make_model
from flygym_amd.controllers import TripodCPG
from flygym_amd.utils.math import Rotation3D
n = 1024
def mk(solver):
    fly = make_model()[0]
    world = C.MixedTerrainWorld()
    world.add_fly(fly, (0, 0, 0.8), Rotation3D("quat", (1, 0, 0, 0)))
    if solver: os.environ["NMF_SOLVER"] = solver
    else: os.environ.pop("NMF_SOLVER", None)
    sim = HIPSimulation(world, n_worlds=n, device=0)
    os.environ.pop("NMF_SOLVER", None)
    return fly, sim
fly, lead = mk("")
others = {k: mk(k)[1] for k in ("nohist", "primal")}
cpg = TripodCPG(fly.get_actuated_jointdofs_order("position"), 1e-4)
table = cpg.targets(n, 2500, device=lead.device, adhesion=(cpg.stance_bins(lead.model, fly), 20.0, 1.0))
ids = lead.replay_ids(fly.name, with_adhesion=True)
lead.set_leg_adhesion_states(fly.name, np.ones((n, 6), dtype=np.float32))
lead.warmup()
keys = ("qpos", "qvel", "ctrl", "qacc_warmstart")
blob = lead.model.to_blob()
cur = 0
rows = []
for cp in range(3):
    for _ in range(6): lead.step_replay(table, ids, cur, 50); cur += 50
    state = {k: lead.field(k).clone() for k in keys}
    for sim in others.values():
        for k in keys: sim.field(k)[:] = state[k]
        sim.step_replay(table, ids, cur, 1)
    lead.step_replay(table, ids, cur, 1); cur += 1
    torch.cuda.synchronize()
    q = {"": lead.field("qacc").cpu().numpy(), **{k: s.field("qacc").cpu().numpy() for k, s in others.items()}}
    st = {"": lead.field("stats").cpu().numpy(), **{k: s.field("stats").cpu().numpy() for k, s in others.items()}}
    spread = np.max([np.abs(q[a] - q[b]).max(axis=1) / np.maximum(np.abs(q['primal']).max(axis=1), 1e4) for a, b in (('', 'primal'), ('nohist', 'primal'))], axis=0)
    for w in range(n):
        if w % 4 and spread[w] < 1e-3: continue          # every fourth world, and every world on which the variants disagree
        o = orc.Oracle(blob, "f64")
        for k in keys: o.arr(k)[:] = state[k][w].cpu().numpy().astype(np.float64)
        o.step_replay(table[w].cpu().numpy(), ids.cpu().numpy(), cur - 1, 1)
        if o.ints()["ncon"] != int(st[""][w, 0]): continue
        a = o.arr("qacc"); sc = max(np.abs(a).max(), 1e4)
        row = (cp, w, int(st[""][w, 0]), o.ints()["solver_iter"]) + tuple((float(np.abs(q[k][w] - a).max() / sc), int(st[k][w, 1])) for k in ("", "nohist", "primal"))
        rows.append(row)
        if row[4][0] > 1e-2 or row[5][0] > 1e-2:
            np.savez(ROOT / "gpurun_out" / f"config5_state_{cp}_{w}.npz", errs=np.array([row[4][0], row[5][0], row[6][0]]), its=np.array([row[4][1], row[5][1], row[6][1]]), qacc_default=q[''][w], qacc_nohist=q['nohist'][w], qacc_primal=q['primal'][w], rows=table[w].cpu().numpy(), cur=cur - 1, **{k: state[k][w].cpu().numpy() for k in keys})
rows.sort(key=lambda r: -r[4][0])
print("checkpoint world ncon oracle_iters | (error, iterations) default / nohist / primal")
for r in rows[:10]: print(r)
for i, k in ((4, "default"), (5, "nohist"), (6, "primal")):
    e = np.array([r[i][0] for r in rows]); print(k, len(e), "states: median %.1e p99 %.1e max %.1e" % (np.median(e), np.quantile(e, 0.99), e.max()))
