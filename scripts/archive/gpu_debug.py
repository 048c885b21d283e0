"""Ad-hoc GPU diagnostics (run through gpurun): HIP engine vs oracle."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "oracle"))
import numpy as np, torch
from flygym_amd import HIPSimulation, make_model
import oracle as orc

fly, world, _ = make_model()
sim = HIPSimulation(world, n_worlds=8, device=0)
blob = sim.model.to_blob()
sim.set_leg_adhesion_states(fly.name, np.ones((8, 6), dtype=np.float32))
o64 = orc.Oracle(blob, "f64"); o64.ctrl[42:] = 1.0
o32 = orc.Oracle(blob, "f32"); o32.ctrl[42:] = 1.0
print("reset qpos err", np.abs(sim.field("qpos").cpu().numpy()[0] - o64.qpos).max())
print("reset segpos err", np.abs(sim.field("seg_xpos").cpu().numpy()[0] - o64.arr("seg_xpos")).max())
for k in range(12):
    sim.step(25); o64.step(25); o32.step(25)
    torch.cuda.synchronize()
    q = sim.field("qpos").cpu().numpy(); v = sim.field("qvel").cpu().numpy()
    st = sim.field("stats").cpu().numpy()[0]
    print(f"step {25*(k+1):4d} z={q[0,2]:.5f} oz={o64.qpos[2]:.5f}  |dq| hip-o64 {np.abs(q[0]-o64.qpos).max():.2e} o32-o64 {np.abs(o32.qpos-o64.qpos).max():.2e}"
          f" |dv| {np.abs(v[0]-o64.qvel).max():.2e} / {np.abs(o32.qvel-o64.qvel).max():.2e} ncon {st[0]:.0f}/{o64.ints()['ncon']} it {st[1]:.0f}/{o64.ints()['solver_iter']} worlds-spread {np.abs(q-q[0]).max():.1e}")
sd = sim.field("sensordata").cpu().numpy()[0].reshape(6, 16)
print("sens hip ", sd[:, :4]); print("sens o64 ", o64.arr("sensordata").reshape(6, 16)[:, :4])
# timing
for n_worlds in (1024, 4096):
    s2 = HIPSimulation(world, n_worlds=n_worlds, device=0)
    s2.set_leg_adhesion_states(fly.name, np.ones((n_worlds, 6), dtype=np.float32))
    s2.step(100); torch.cuda.synchronize()
    t = time.time(); s2.step(200); torch.cuda.synchronize(); dt = time.time() - t
    print(f"n_worlds {n_worlds}: {200*n_worlds/dt:.3e} env-steps/s  ({dt/200*1e6:.1f} us/step)")
