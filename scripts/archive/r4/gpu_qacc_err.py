"""One-step acceleration error of the engine against the float64 oracle on contact-rich states (diagnostic, run through gpurun):
usage: NMF_HIP_LIB=build/libnmf_<variant>.so python scripts/r4/gpu_qacc_err.py [joint preset]
Also the contact sensors of the same step: net force and force-weighted centroid per leg (what a near-degenerate split of
the load between two contacts of one leg shows up in)."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "oracle"))
import numpy as np, torch
import oracle as orc
from flygym_amd import HIPSimulation, make_model
from flygym_amd.controllers import TripodCPG
fly, world, _ = make_model(joints_preset=sys.argv[1] if len(sys.argv) > 1 else "legs_only")
n = 256
sim = HIPSimulation(world, n_worlds=n, device=0)
cpg = TripodCPG(fly.get_actuated_jointdofs_order("position"), 1e-4)
table = cpg.targets(n, 2500, device=sim.device)
ids = sim.replay_ids(fly.name)
sim.set_leg_adhesion_states(fly.name, np.ones((n, 6), dtype=np.float32))
sim.warmup(); sim.step_replay(table, ids, 0, 900)
errs, ferr, its, cerr = [], [], [], []
blob = sim.model.to_blob()
for rep in range(3):
    sim.step_replay(table, ids, 900 + 40 * rep, 39)
    torch.cuda.synchronize()
    state = {k: sim.field(k).cpu().numpy().astype(np.float64) for k in ("qpos", "qvel", "ctrl", "qacc_warmstart")}
    sim.step_replay(table, ids, 900 + 40 * rep + 39, 1)
    torch.cuda.synchronize()
    qacc = sim.field("qacc").cpu().numpy(); st = sim.field("stats").cpu().numpy()
    sd = sim.field("sensordata").cpu().numpy().reshape(n, 6, 16)
    for w in range(0, n, 4):
        o = orc.Oracle(blob, "f64")
        o.qpos[:] = state["qpos"][w]; o.qvel[:] = state["qvel"][w]; o.arr("qacc_warmstart")[:] = state["qacc_warmstart"][w]
        o.ctrl[:] = state["ctrl"][w]
        o.step_replay(table[w].cpu().numpy(), ids.cpu().numpy(), 900 + 40 * rep + 39, 1)
        if o.ints()["ncon"] != int(st[w, 0]): continue
        a = o.arr("qacc")
        errs.append(np.abs(qacc[w] - a).max() / np.abs(a).max()); its.append(st[w, 1])
        so = o.arr("sensordata").reshape(6, 16)
        ferr.append(np.abs(sd[w][:, 1:4] - so[:, 1:4]).max() / max(np.abs(so[:, 1:4]).max(), 1e-30))
        cerr.append(np.abs(sd[w][:, 7:10] - so[:, 7:10]).max())
errs = np.array(errs)
ferr, cerr = np.array(ferr), np.array(cerr)
print(f"sensor net force error / max median {np.median(ferr):.2e} p90 {np.quantile(ferr, 0.9):.2e} max {ferr.max():.2e};  centroid error [mm] median {np.median(cerr):.2e} p90 {np.quantile(cerr, 0.9):.2e} max {cerr.max():.2e}")
print(f"{len(errs)} states: qacc error / max|qacc| median {np.median(errs):.2e}  p90 {np.quantile(errs, 0.9):.2e}  max {errs.max():.2e}  mean iterations {np.mean(its):.2f}")
