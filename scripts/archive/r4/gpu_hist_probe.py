"""Which steps of the history-started contact-space solve end away from the oracle's optimum?  (diagnostic, run through gpurun)
usage: python scripts/r4/gpu_hist_probe.py [joint preset]   — steps n walkers, compares every world's one-step qacc with the
float64 oracle at several checkpoints and prints the worst ones with their contact counts and iteration counts, then the
same states stepped by the other solver variants."""
import os, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "oracle"))
import numpy as np, torch
import oracle as orc
from flygym_amd import HIPSimulation, make_model
from flygym_amd.controllers import TripodCPG
preset = sys.argv[1] if len(sys.argv) > 1 else "legs_only"
n = 256
def mk(solver):
    fly, world, _ = make_model(joints_preset=preset)
    if solver: os.environ["NMF_SOLVER"] = solver
    else: os.environ.pop("NMF_SOLVER", None)
    sim = HIPSimulation(world, n_worlds=n, device=0)
    os.environ.pop("NMF_SOLVER", None)
    return fly, sim
fly, lead = mk("")
others = {k: mk(k)[1] for k in ("nohist", "primal")}
table = TripodCPG(fly.get_actuated_jointdofs_order("position"), 1e-4).targets(n, 2500, device=lead.device)
ids = lead.replay_ids(fly.name)
lead.set_leg_adhesion_states(fly.name, np.ones((n, 6), dtype=np.float32))
lead.warmup(); lead.step_replay(table, ids, 0, 850)
keys = ("qpos", "qvel", "ctrl", "qacc_warmstart")
blob = lead.model.to_blob()
cur = 850
rows = []
for cp in range(8):
    lead.step_replay(table, ids, cur, 23); cur += 23
    state = {k: lead.field(k).clone() for k in keys}
    for sim in others.values():
        for k in keys: sim.field(k)[:] = state[k]
        sim.step_replay(table, ids, cur, 1)
    lead.step_replay(table, ids, cur, 1); cur += 1
    torch.cuda.synchronize()
    q = {"": lead.field("qacc").cpu().numpy(), **{k: s.field("qacc").cpu().numpy() for k, s in others.items()}}
    st = {"": lead.field("stats").cpu().numpy(), **{k: s.field("stats").cpu().numpy() for k, s in others.items()}}
    for w in range(n):
        o = orc.Oracle(blob, "f64")
        for k in keys: o.arr(k)[:] = state[k][w].cpu().numpy().astype(np.float64)
        o.step_replay(table[w].cpu().numpy(), ids.cpu().numpy(), cur - 1, 1)
        if o.ints()["ncon"] != int(st[""][w, 0]): continue
        a = o.arr("qacc"); sc = np.abs(a).max()
        rows.append((cp, w, int(st[""][w, 0]), o.ints()["solver_iter"]) + tuple((float(np.abs(q[k][w] - a).max() / sc), int(st[k][w, 1])) for k in ("", "nohist", "primal")))
rows.sort(key=lambda r: -r[4][0])
print("checkpoint world ncon oracle_iters | (error, iterations) default / nohist / primal")
for r in rows[:12]: print(r)
dualrows = [r for r in rows if r[2] <= (10 if preset == "all_biological" else 12)]
e = np.array([r[4][0] for r in rows]); print(len(rows), "states; default: median %.1e p99 %.1e max %.1e" % (np.median(e), np.quantile(e, 0.99), e.max()), "mean iterations", np.mean([r[4][1] for r in rows]))
e = np.array([r[4][0] for r in dualrows]); print(len(dualrows), "of them on the contact-space path; default: median %.1e p99 %.1e max %.1e" % (np.median(e), np.quantile(e, 0.99), e.max()), "mean iterations", np.mean([r[4][1] for r in dualrows]), "primal on the same: max %.1e" % max(r[6][0] for r in dualrows))
print("worst on the contact-space path:", sorted(dualrows, key=lambda r: -r[4][0])[:5])
for i, k in ((5, "nohist"), (6, "primal")):
    e = np.array([r[i][0] for r in rows]); print(k, "median %.1e p99 %.1e max %.1e" % (np.median(e), np.quantile(e, 0.99), e.max()))
