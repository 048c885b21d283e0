"""Where does a flat-ground replay rollout leave the float64 oracle?  Per step: engine's own state -> oracle -> compare the
step's acceleration (diagnostic, run through gpurun)."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "oracle"))
import numpy as np, torch
import oracle as orc
from flygym_amd import HIPSimulation, make_model
from flygym_amd.compose import ActuatorType
from flygym_amd.replay import ReplayTargetData
fly, world, _ = make_model()
n = 4096
order = fly.get_actuated_jointdofs_order(ActuatorType.POSITION)
table_np = ReplayTargetData(1e-4, order).make_target_angles_all_worlds(n, 1000)
sim = HIPSimulation(world, n_worlds=n, device=0)
ids = sim.replay_ids(fly.name)
sim.set_leg_adhesion_states(fly.name, np.ones((n, 6), dtype=np.float32))
sim.step(300)
tab = torch.as_tensor(table_np, device=sim.device)
blob = sim.model.to_blob()
base = orc.Oracle(blob, "f64"); base.ctrl[42:] = 1.0; base.step(300)
picks = np.random.default_rng(2).choice(n, size=32, replace=False)
refs = {int(w): base.clone_data() for w in picks}
keys = ("qpos", "qvel", "ctrl", "qacc_warmstart")
worst = {}
for k in range(150):
    before = {kk: sim.field(kk)[torch.as_tensor(picks, device=sim.device)].cpu().numpy().astype(np.float64) for kk in keys}
    sim.step_replay(tab, ids, k, 1)
    qacc = sim.field("qacc")[torch.as_tensor(picks, device=sim.device)].cpu().numpy()
    st = sim.field("stats")[torch.as_tensor(picks, device=sim.device)].cpu().numpy()
    geom = sim.field("contact_geom")[torch.as_tensor(picks, device=sim.device)].cpu().numpy()
    for j, w in enumerate(picks):
        w = int(w)
        refs[w].step_replay(table_np[w], np.arange(42), k, 1)          # the free-running oracle
        r = orc.Oracle(blob, "f64")
        for kk in keys: r.arr(kk)[:] = before[kk][j]
        r.step_replay(table_np[w], np.arange(42), k, 1)               # one step from the engine's own state
        a = r.arr("qacc"); nc = int(st[j, 0])
        same = geom[j, :nc].astype(int).tolist() == r.ints()["con_geom"]
        err = np.abs(qacc[j] - a).max() / np.abs(a).max()
        drift = np.abs(sim.field("qpos")[w].cpu().numpy() - refs[w].qpos).max()
        if w in (2278, 1363, 3590) and (k % 10 == 0 or err > 3e-4): print(w, "step", k, f"drift {drift:.1e} step err {err:.1e} same {same} ncon {nc} iters {int(st[j,1])}/{r.ints()['solver_iter']}")
        if (err > 2e-3 or not same) and w % 20 in (3, 5, 12, 13, 17, 8, 0, 1, 2, 4, 6, 7, 9, 10, 11, 14, 15, 16, 18, 19):
            worst.setdefault(w, []).append((k, err, same, nc, int(st[j, 1]), r.ints()["solver_iter"], drift))
for w, ev in worst.items():
    print("world", w, "partition", w % 20, "events", [(k, f"{e:.1e}", s, nc, it, oi, f"{d:.1e}") for k, e, s, nc, it, oi, d in ev[:6]])
fin = {int(w): np.abs(sim.field("qpos")[int(w)].cpu().numpy() - refs[int(w)].qpos).max() for w in picks}
print("final drift per world", {w: f"{v:.1e}" for w, v in fin.items() if v > 5e-5})
