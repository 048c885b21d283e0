"""Diagnostic (run through gpurun): belly-contact scenario, engine re-synchronised to the float32 oracle every n steps."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "oracle"))
import numpy as np, torch
from flygym_amd import HIPSimulation, make_model
from flygym_amd.compose import ActuatorType
import oracle as orc
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
fly, world, _ = make_model()
a = HIPSimulation(world, n_worlds=1, device=0)
o = orc.Oracle(a.model.to_blob(), "f32")
o.ctrl[:42] = 0; o.qpos[2] = 0.3
worst = []
for k in range(660 // n):
    for name, v in (("qpos", o.qpos), ("qvel", o.qvel), ("ctrl", o.ctrl), ("qacc_warmstart", o.arr("qacc_warmstart"))):
        a.field(name)[:] = torch.as_tensor(np.asarray(v), dtype=torch.float32, device=a.device)
    a.step(n); o.step(n)
    q = a.field("qpos").cpu().numpy()[0]; st = a.field("stats").cpu().numpy()[0]
    e = np.abs(q - o.qpos).max()
    worst.append(e)
    if e > 2e-6 or int(st[0]) != o.ints()["ncon"]:
        print(f"step {n*(k+1)}: err {e:.2e} ncon {int(st[0])}/{o.ints()['ncon']} iters {int(st[1])}/{o.ints()['solver_iter']}")
print("segments", len(worst), "max", max(worst), "median", float(np.median(worst)))
