"""Diagnostic (gpurun): how predictable is a world's cost from launch to launch, and how balanced would a static,
cost-sorted pairing of worlds be?  Runs 20-step launches under NMF_SCHED (paired / chunks / plain) and prints statistics."""
import os, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import numpy as np, torch
from flygym_amd import HIPSimulation, make_model
from flygym_amd.compose import ActuatorType
from flygym_amd.controllers import TripodCPG
n, spl = 4096, int(os.environ.get("SPL", "20"))
fly, world, _ = make_model()
order = fly.get_actuated_jointdofs_order(ActuatorType.POSITION)
sim = HIPSimulation(world, n_worlds=n, device=0)
table = TripodCPG(order, sim.timestep).targets(n, 2500, device=sim.device)
ids = sim.replay_ids(fly.name)
sim.set_leg_adhesion_states(fly.name, np.ones((n, 6), dtype=np.float32))
sim.step(500)
cur = 0
for _ in range(17):
    sim.step_replay(table, ids, cur, 50); cur += 50
costs, its = [], []
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(12)]
for k in range(12):
    s0 = sim.field("stats_sum").clone()
    ev[k][0].record(); sim.step_replay(table, ids, cur, spl); ev[k][1].record(); cur += spl
    torch.cuda.synchronize()
    costs.append(sim.field("cost")[:, 0].cpu().numpy().copy())
    d = (sim.field("stats_sum") - s0).cpu().numpy()
    its.append(d[:, 2].copy())
c = np.array(costs); it = np.array(its)
ms = [a.elapsed_time(b) for a, b in ev]
clk = sim.shader_clock_hz()
print("sched", os.environ.get("NMF_SCHED"), "spl", spl, "ms per launch", np.round(ms[2:], 3), "clock GHz", clk / 1e9)
for k in (10, 11):
    a, b = c[k - 1], c[k]
    print(f"launch {k}: cost mean {b.mean():.0f} min {b.min():.0f} max {b.max():.0f} cycles; corr(prev) {np.corrcoef(a, b)[0, 1]:.3f}; "
          f"corr(cost, newton iters) {np.corrcoef(b, it[k])[0, 1]:.3f}; corr(iters, prev iters) {np.corrcoef(it[k], it[k-1])[0,1]:.3f}")
    o = np.argsort(-a)                         # costliest first by the previous launch
    pair_pred = a[o[:n // 2]] + a[o[::-1][:n // 2]]
    pair_act = b[o[:n // 2]] + b[o[::-1][:n // 2]]
    print(f"   pair sums by previous cost: predicted max/mean {pair_pred.max() / pair_pred.mean():.3f}, actual max/mean {pair_act.max() / pair_act.mean():.3f}; "
          f"ideal launch = mean pair sum {pair_act.mean() / clk * 1e3:.3f} ms, max pair sum {pair_act.max() / clk * 1e3:.3f} ms")
    oi = np.argsort(-it[k - 1])
    pit = it[k][oi[:n // 2]] + it[k][oi[::-1][:n // 2]]
    print(f"   pairing by previous Newton iterations: iteration pair sums max/mean {pit.max() / pit.mean():.3f}; cost pair sums max/mean "
          f"{(b[oi[:n // 2]] + b[oi[::-1][:n // 2]]).max() / (b[oi[:n // 2]] + b[oi[::-1][:n // 2]]).mean():.3f}")
np.save(ROOT / "gpurun_out" / f"costpair_{os.environ.get('NMF_SCHED', 'auto')}.npy", np.stack([c, it]))
