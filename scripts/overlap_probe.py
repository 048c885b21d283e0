"""Round 6, verdict item 3(ii): does the eye renderer of vision tick k pay beside the physics of tick k + 1?

TIMING ONLY (the renderer reads whatever poses the running physics has published: the readings of this probe are not used).
For every residency of the stepping kernel (flies per CU: the kernel's own 8, then 7, 6, 5, 4 through
nmf_batch_options.flies_per_cu = idle LDS per workgroup) it measures, per vision tick of 20 steps on 4096 flies:
  serial   physics, then eyes, one stream (what bench.py --vision render does)
  overlap  physics of tick k + 1 on stream A while the eyes of tick k run on stream B (events order eyes(k) after
           physics(k) and physics(k + 2) after eyes(k): the pose snapshot is double-buffered in the real thing)
and the two kernels alone.  usage: python scripts/overlap_probe.py [--worlds 4096] [--ticks 40]"""
import argparse, json, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import numpy as np, torch
from flygym_amd import HIPSimulation, make_model
from flygym_amd.controllers import TripodCPG
from flygym_amd.vision import EyeRenderer, Scene

ap = argparse.ArgumentParser()
ap.add_argument("--worlds", type=int, default=4096)
ap.add_argument("--ticks", type=int, default=40)
ap.add_argument("--every", type=int, default=20)
ap.add_argument("--residency", default="0,7,6,5,4")
ap.add_argument("--eye-rays", type=int, default=0)
args = ap.parse_args()
n, dev = args.worlds, torch.device("cuda", 0)
rows = []
for res in [int(x) for x in args.residency.split(",")]:
    fly, world, _ = make_model()
    sim = HIPSimulation(world, n_worlds=n, device=0, _options=dict(flies_per_cu=res) if res else None)
    info = sim.batch_info()
    eyes = EyeRenderer(sim, fly.name, Scene(spheres=[(8.0, 3.0, 1.5, 1.0)], sphere_rgb=[(0.05, 0.05, 0.05)]), rays_per_ommatidium=args.eye_rays) \
        if args.eye_rays else EyeRenderer(sim, fly.name, Scene(spheres=[(8.0, 3.0, 1.5, 1.0)], sphere_rgb=[(0.05, 0.05, 0.05)]))
    out = torch.zeros((n, 2, eyes.retina.num_ommatidia, 2), device=dev)
    sim.set_leg_adhesion_states(fly.name, np.ones((n, 6), dtype=np.float32))
    sim.warmup()
    table = TripodCPG(fly.get_actuated_jointdofs_order("position"), sim.timestep).targets(n, 2500, device=dev)
    ids = sim.replay_ids(fly.name)
    sim.step_replay(table, ids, 0, 200)
    eyes.render_into(out); torch.cuda.synchronize()
    k0 = [10]

    def physics():
        sim.step_replay(table, ids, (k0[0] * args.every) % 2000, args.every); k0[0] += 1

    def timed(fn, reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3

    for _ in range(5): physics()
    t_phys = timed(physics, args.ticks)
    t_eyes = timed(lambda: eyes.render_into(out), 10)
    t_serial = timed(lambda: (physics(), eyes.render_into(out)), args.ticks)
    sA, sB = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    ev_p = [torch.cuda.Event() for _ in range(args.ticks + 4)]
    ev_e = [torch.cuda.Event() for _ in range(args.ticks + 4)]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(args.ticks):
        with torch.cuda.stream(sA):
            if k >= 2: sA.wait_event(ev_e[k - 2])
            physics(); ev_p[k].record(sA)
        with torch.cuda.stream(sB):
            sB.wait_event(ev_p[k])
            eyes.render_into(out); ev_e[k].record(sB)
    torch.cuda.synchronize(); t_over = (time.perf_counter() - t0) / args.ticks * 1e3
    row = dict(flies_per_cu=info["flies_per_cu"], resident=info["resident_workgroups"], physics_ms=round(t_phys, 4), eyes_ms=round(t_eyes, 4),
               serial_tick_ms=round(t_serial, 4), overlapped_tick_ms=round(t_over, 4),
               serial_M=round(n * args.every / t_serial / 1e3, 2), overlapped_M=round(n * args.every / t_over / 1e3, 2))
    print(json.dumps(row), flush=True)
    rows.append(row)
    del eyes, sim
