"""sha256 of the state after a fixed rollout (run through gpurun with NMF_HIP_LIB=<variant>): two libraries whose arithmetic is the
same print the same hash.  usage: python scripts/state_hash.py [worlds] [terrain]"""
import hashlib, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import numpy as np, torch
from flygym_amd import HIPSimulation, make_model, _native
from flygym_amd.controllers import TripodCPG

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
fly, world, _ = make_model()
if len(sys.argv) > 2 and sys.argv[2] != "flat":
    import flygym_amd.compose as C
    from flygym_amd.utils.math import Rotation3D
    world = {"gapped": C.GappedTerrainWorld, "blocks": C.BlocksTerrainWorld, "mixed": C.MixedTerrainWorld}[sys.argv[2]]()
    world.add_fly(fly, (0, 0, 0.8), Rotation3D("quat", (1, 0, 0, 0)))
sim = HIPSimulation(world, n_worlds=n, device=0)
sim.set_leg_adhesion_states(fly.name, np.ones((n, 6), dtype=np.float32))
sim.warmup()
table = TripodCPG(fly.get_actuated_jointdofs_order("position"), sim.timestep).targets(n, 2500, device=sim.device)
ids = sim.replay_ids(fly.name)
for k in range(20):
    sim.step_replay(table, ids, 50 * k, 50)
torch.cuda.synchronize()
h = hashlib.sha256()
for f in ("qpos", "qvel", "qacc", "qacc_warmstart", "sensordata"):
    h.update(sim.field(f).cpu().numpy().tobytes())
print(_native.LIB_PATH.name, n, h.hexdigest()[:16], sim.get_solver_exits())
