"""A tour of the engine on one MI355X: 1024 flies walk with a tripod gait, smell three odor sources and see a dark
sphere through their compound eyes.  Run from the repo root:  python examples/walk_and_see.py

The first half is the reference's own usage pattern (flygym tutorials / `flygym_demo.benchmark`):
`make_model()`, a simulation object, `set_leg_adhesion_states`, `warmup`, `set_actuator_inputs` + `step`, getters.
The second half shows what the batched engine adds: a device-resident control table (`step_replay`), gait-driven
adhesion, per-world resets and the fused eye renderer.
"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
import torch

from flygym_amd import HIPSimulation, make_model
from flygym_amd.compose import ActuatorType
from flygym_amd.controllers import TripodCPG
from flygym_amd.sensors import OdorSensors
from flygym_amd.vision import EyeRenderer, Scene

n = 1024
fly, world, _ = make_model()                       # LEGS_ONLY fly on flat ground, 42 position actuators, leg adhesion
sim = HIPSimulation(world, n_worlds=n)
sim.set_leg_adhesion_states(fly.name, np.ones((n, 6), dtype=np.float32))
sim.warmup()                                       # 500 steps: the flies settle on the ground

# --- the reference's per-step loop: write targets, step, read state
order = fly.get_actuated_jointdofs_order(ActuatorType.POSITION)
cpg = TripodCPG(order, sim.timestep)
targets = cpg.targets(n, 100, device=sim.device)   # (n, 100, 42) on the GPU
for k in range(100):
    sim.set_actuator_inputs(fly.name, ActuatorType.POSITION, targets[:, k])
    sim.step()
angles = sim.get_joint_angles(fly.name)            # torch (n, 66) on the GPU
print(f"t = {sim.time * 1e3:.1f} ms, joint angle range {angles.min().item():+.2f} .. {angles.max().item():+.2f} rad")

# --- device-resident loop: the whole control table lives on the GPU, 50 steps per launch, adhesion follows the gait
stance = cpg.stance_bins(sim.model, fly)
table = cpg.targets(n, 2500, device=sim.device, adhesion=(stance, 20.0, 1.0))        # 42 joint targets + 6 adhesion
ids = sim.replay_ids(fly.name, with_adhesion=True)
odor = OdorSensors(sim, fly.name, source_positions=[(15, 0, 1.5), (-5, 12, 1.5), (4, -9, 1.5)],
                   peak_intensities=[(1.0, 0.0), (0.0, 1.0), (0.5, 0.5)])
eyes = EyeRenderer(sim, fly.name, Scene(spheres=[(10.0, 3.0, 1.5, 1.0)], sphere_rgb=[(0.05, 0.05, 0.05)]))  # sees the ground, the sphere and its own legs
x0 = sim.get_body_positions(fly.name)[:, 0, 0].clone()
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ev0.record()
for tick in range(100):                            # 100 control ticks of 50 steps = 0.5 s of walking
    sim.step_replay(table, ids, 100 + 50 * tick, 50)
    smell = odor.get_odor_intensities()            # (n, 2, 4)
    sight = eyes.render()                          # (n, 2, 721, 2) ommatidia readings
    fallen = sim.get_body_positions(fly.name)[:, 0, 2] < 0.3
    if fallen.any():
        sim.reset_worlds(fallen)                   # episode reset of just those worlds
ev1.record(); torch.cuda.synchronize()
dx = (sim.get_body_positions(fly.name)[:, 0, 0] - x0).mean().item()
active, force, *_ = sim.get_ground_contact_info(fly.name)
print(f"walked {dx:.2f} mm on average in 0.5 s; {active.sum(1).float().mean().item():.1f} legs on the ground; "
      f"odor at the left antenna {smell[:, :, 2].mean().item():.4f}; mean ommatidia reading {sight.mean().item():.3f}")
print(f"{n * 5000 / (ev0.elapsed_time(ev1) * 1e-3):.3e} env-steps/s including smell and sight every 50 steps")
