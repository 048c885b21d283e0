"""BASELINE config 3: 4096 flies with vision on — per 500 Hz vision tick (every 20 physics steps) two 512x450
raw eye frames per fly are resampled to 2 x 721 x 2 ommatidia readings.  Raw frames are synthetic (no renderer in
scope): seeded uint8 noise over a checker floor, one distinct frame pair per fly.  Prints one JSON line with the
resample kernel's HBM roofline and the combined physics + vision throughput."""
import argparse, json, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import numpy as np, torch
from flygym_amd import HIPSimulation, make_model
from flygym_amd.compose import ActuatorType
from flygym_amd.replay import ReplayTargetData
from flygym_amd.sensors import Retina, RAW_IMG_HEIGHT as H, RAW_IMG_WIDTH as W

ap = argparse.ArgumentParser()
ap.add_argument("--worlds", type=int, default=4096)
ap.add_argument("--steps", type=int, default=1000)
ap.add_argument("--vision-every", type=int, default=20)
ap.add_argument("--render", action="store_true", help="render the eye views on the GPU (fused with the resample) instead of resampling pre-made frames")
args = ap.parse_args()
n, dev = args.worlds, torch.device("cuda", 0)
fly, world, _ = make_model()
sim = HIPSimulation(world, n_worlds=n, device=0)
retina = Retina()
g = torch.Generator(device=dev); g.manual_seed(0)
if args.render:
    from flygym_amd.vision import EyeRenderer, Scene
    eyes = EyeRenderer(sim, fly.name, Scene(spheres=[(8.0, 3.0, 1.5, 1.0)], sphere_rgb=[(0.05, 0.05, 0.05)]), retina)
    frames = None
    see = lambda: eyes.render()
else:
    frames = torch.randint(0, 256, (n, 2, H, W, 3), dtype=torch.uint8, device=dev, generator=g)
    yy, xx = torch.meshgrid(torch.arange(H, device=dev), torch.arange(W, device=dev), indexing="ij")
    frames[:, :, ((yy // 32 + xx // 32) % 2 == 0) & (yy > H // 2)] //= 4          # darker checker floor
    see = lambda: retina.raw_image_to_hex_pxls(frames)
order = fly.get_actuated_jointdofs_order(ActuatorType.POSITION)
table = torch.as_tensor(ReplayTargetData(sim.timestep, order).make_target_angles_all_worlds(n, 1000), device=dev)
ids = sim._ids_by_fly[fly.name]["actuators"][ActuatorType.POSITION]
sim.set_leg_adhesion_states(fly.name, np.ones((n, 6), dtype=np.float32))
sim.step(500)
out = see(); torch.cuda.synchronize()
# resample alone
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = 10
e0.record()
for _ in range(reps): out = see()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
bytes_in = frames.numel() if frames is not None else 0; bytes_out = out.numel() * 4
# physics + vision
ticks = args.steps // args.vision_every
torch.cuda.synchronize(); t0 = time.perf_counter()
for k in range(ticks):
    sim.step_replay(table, ids, k * args.vision_every, args.vision_every)
    out = see()
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(json.dumps({
    "metric": "env-steps/sec, 4096 flies with vision on (2 x 721-ommatidia resample per 20 steps), 1x MI355X",
    "value": n * ticks * args.vision_every / dt, "unit": "env-steps/s", "n_gpus": 1, "steps": ticks * args.vision_every,
    "data": ("eye views ray-cast on the GPU (checker ground, sky, one sphere), fused with the resample" if args.render else
             "synthetic raw eye frames (seeded noise over a checker floor)") + ", replay-walking physics",
    "config": {"workload": "BASELINE config 3", "worlds": n, "vision_every_steps": args.vision_every,
               "frames_per_tick": 2 * n, "frame_bytes": H * W * 3},
    "rays_per_s": (2 * n * H * W / (ms * 1e-3)) if args.render else None,
    "roofline": {"bound": "hbm", "kernel": "nmf_eye_kernel" if args.render else "nmf_retina_stream_kernel", "achieved": (bytes_in + bytes_out) / (ms * 1e-3) / 1e9, "peak": 8000.0,
                 "unit": "GB/s", "frac": (bytes_in + bytes_out) / (ms * 1e-3) / 1e9 / 8000.0, "kernel_ms": ms,
                 "algorithmic_bytes_per_launch": bytes_in + bytes_out},
}))
