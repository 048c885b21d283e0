"""Soak of the stepping kernel (run through gpurun): N worlds x many steps of CPG walking on several worlds / skeletons;
checks that the state stays finite, no world overflows its contact list, every world keeps walking (forward speed) and
the solver's iteration counts stay bounded.  usage: python scripts/gpu_soak.py [steps] [all]   (round 5 log of record: profiles/r5_soak.txt = `gpu_soak.py 50000 all`: 205 M env-steps per workload)"""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import numpy as np, torch
import flygym_amd.compose as C
from flygym_amd import HIPSimulation, make_model
from flygym_amd.controllers import TripodCPG
from flygym_amd.utils.math import Rotation3D
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
n = 4096
WORKLOADS = (("flat LEGS_ONLY", None, "legs_only", 0.0), ("mixed + 20x gait adhesion", "MixedTerrainWorld", "legs_only", 20.0),
             ("blocks", "BlocksTerrainWorld", "legs_only", 0.0), ("flat ALL_BIOLOGICAL", None, "all_biological", 0.0),
             # the wider set (second argument "all")
             ("gapped", "GappedTerrainWorld", "legs_only", 0.0), ("flat LEGS_ACTIVE_ONLY", None, "legs_active_only", 0.0),
             ("tethered", "TetheredWorld", "legs_only", 0.0), ("mixed ALL_BIOLOGICAL", "MixedTerrainWorld", "all_biological", 0.0),
             ("blocks + 20x gait adhesion", "BlocksTerrainWorld", "legs_only", 20.0), ("flat ALL_POSSIBLE", None, "all_possible", 0.0))
for name, world_cls, preset, adhesion in (WORKLOADS if len(sys.argv) > 2 and sys.argv[2] == "all" else WORKLOADS[:4]):
    fly, world, _ = make_model(joints_preset=preset)
    if world_cls:
        world = getattr(C, world_cls)()
        world.add_fly(fly, (0, 0, 1.5 if world_cls == "TetheredWorld" else 0.8), Rotation3D("quat", (1, 0, 0, 0)))
    sim = HIPSimulation(world, n_worlds=n, device=0)
    if name == WORKLOADS[0][0]: print("batch_info", sim.batch_info())
    cpg = TripodCPG(fly.get_actuated_jointdofs_order("position"), 1e-4)
    table = cpg.targets(n, 2500, device=sim.device, adhesion=(cpg.stance_bins(sim.model, fly), adhesion, 1.0) if adhesion else None)
    ids = sim.replay_ids(fly.name, with_adhesion=bool(adhesion))
    sim.set_leg_adhesion_states(fly.name, np.ones((n, 6), dtype=np.float32))
    sim.warmup()
    x0 = sim.field("qpos")[:, 0].clone()
    t0 = time.time(); worst_it = 0; bad = 0
    for k in range(0, steps, 50):
        sim.step_replay(table, ids, k, 50)
        if k % 1000 == 0:
            st = sim.field("stats")
            worst_it = max(worst_it, int(st[:, 1].max().item()))
            bad += int((~torch.isfinite(sim.field("qpos")).all(dim=1)).sum().item())
    torch.cuda.synchronize()
    ss = sim.field("stats_sum").to(torch.int64)
    q = sim.field("qpos")
    dx = (q[:, 0] - x0).cpu().numpy()          # (the open-loop tripod gait walks a circle of ~4 mm radius — in the oracle too — so this is a range check, not a distance)
    speed = sim.field("qvel")[:, :2].norm(dim=1).cpu().numpy()
    print(f"{name:28s} {steps} steps x {n} worlds in {time.time() - t0:5.1f} s: finite {bool(torch.isfinite(q).all())} (non-finite samples {bad}), "
          f"overflow steps {int(ss[:, 3].sum())}, contacts/step {ss[:, 1].sum().item() / ss[:, 0].sum().item():.2f}, iterations/step {ss[:, 2].sum().item() / ss[:, 0].sum().item():.2f} "
          f"(max seen at a sample {worst_it}), solve reports per million steps { {k: round(v * 1e6 / max(sim.get_solver_exits()['steps'], 1), 2) for k, v in sim.get_solver_exits().items() if k != 'steps' and v} }, x displacement median {np.median(dx):.2f} mm (min {dx.min():.2f}, max {dx.max():.2f}), ground speed median {np.median(speed):.1f} mm/s (max {speed.max():.1f}), body height median {float(q[:, 2].median()):.3f}")
