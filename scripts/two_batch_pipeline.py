"""Experiment (gpurun): the 4096 flies of BASELINE config 2 as TWO 2048-fly batches stepped on two HIP streams in 20-step
launches.  Each launch is a plain one (2048 worlds = every resident wave, no chunks, no tickets, no hand-off); the next
launch of the other batch fills the slots the fast worlds free — the ragged end of one launch overlaps the start of the
next.  What a closed-loop user with two environment groups (act on one while the other steps) would run."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import numpy as np, torch
from flygym_amd import HIPSimulation, make_model
from flygym_amd.compose import ActuatorType
from flygym_amd.controllers import TripodCPG

spl = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n_half, total = 2048, 4096
sims, tables, ids, streams = [], [], [], []
for h in range(2):
    fly, world, _ = make_model()
    sim = HIPSimulation(world, n_worlds=n_half, device=0)
    cpg = TripodCPG(fly.get_actuated_jointdofs_order(ActuatorType.POSITION), sim.timestep)
    tables.append(cpg.targets(n_half, 2500, device=sim.device, first_world=h * n_half, total_worlds=total))
    ids.append(sim.replay_ids(fly.name))
    sim.set_leg_adhesion_states(fly.name, np.ones((n_half, 6), dtype=np.float32))
    sim.step(500)
    for k in range(17):
        sim.step_replay(tables[-1], ids[-1], 50 * k, 50)
    sims.append(sim); streams.append(torch.cuda.Stream())
torch.cuda.synchronize()
s0 = [s.field("stats_sum").clone() for s in sims]
cur, regions = 850, 1000 // spl
for s in streams:
    s.wait_stream(torch.cuda.current_stream())
torch.cuda.synchronize()
t0 = time.perf_counter()
for r in range(regions):
    for h in range(2):
        with torch.cuda.stream(streams[h]):
            sims[h].step_replay(tables[h], ids[h], cur, spl)
    cur += spl
torch.cuda.synchronize()
dt = time.perf_counter() - t0
steps = sum(int((s.field("stats_sum") - z)[:, 0].sum().item()) for s, z in zip(sims, s0))
con = sum(float((s.field("stats_sum") - z)[:, 1].double().sum().item()) for s, z in zip(sims, s0)) / steps
assert steps == total * regions * spl
print(f"two batches of {n_half} on two streams, {spl}-step launches: {steps / dt / 1e6:.2f} M env-steps/s ({dt / regions * 1e3:.3f} ms per tick of both batches), mean contacts {con:.2f}")
# the same on one stream (launches serialised): what the overlap is worth
t0 = time.perf_counter()
for r in range(regions):
    for h in range(2):
        sims[h].step_replay(tables[h], ids[h], cur, spl)
    cur += spl
torch.cuda.synchronize()
dt1 = time.perf_counter() - t0
print(f"  same launches on ONE stream: {total * regions * spl / dt1 / 1e6:.2f} M env-steps/s")
