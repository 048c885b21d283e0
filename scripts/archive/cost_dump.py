"""Diagnostic (gpurun): per-world cost profile of a launch for the cpg and replay workloads -> gpurun_out/cost_*.npy"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import numpy as np, torch
from flygym_amd import HIPSimulation, make_model
from flygym_amd.compose import ActuatorType
from flygym_amd.controllers import TripodCPG
from flygym_amd.replay import ReplayTargetData
n = 4096
fly, world, _ = make_model()
order = fly.get_actuated_jointdofs_order(ActuatorType.POSITION)
for wl in ("cpg", "replay"):
    sim = HIPSimulation(world, n_worlds=n, device=0)
    if wl == "cpg":
        table = TripodCPG(order, sim.timestep).targets(n, 2500, device=sim.device); ts = 2500
    else:
        table = torch.as_tensor(ReplayTargetData(sim.timestep, order).make_target_angles_all_worlds(n, 1000), device=sim.device); ts = 1000
    ids = sim.replay_ids(fly.name)
    sim.set_leg_adhesion_states(fly.name, np.ones((n, 6), dtype=np.float32))
    sim.step(500)
    costs = []
    for k in range(12):
        sim.step_replay(table, ids, 50 * k, 50)
        costs.append(sim.field("cost")[:, 0].cpu().numpy().copy())
    c = np.array(costs)
    np.save(ROOT / "gpurun_out" / f"cost_{wl}.npy", c)
    last = c[-1]
    rough = np.abs(np.diff(last)).sum(); spread = np.abs(last - last.mean()).sum()
    print(wl, "mean", last.mean(), "min", last.min(), "max", last.max(), "rough/spread", rough / spread,
          "corr with previous launch", np.corrcoef(c[-1], c[-2])[0, 1])
