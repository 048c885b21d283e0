"""Per-stage cycle breakdown of the step kernel (diagnostic build, run through gpurun)."""
import ctypes, subprocess, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import numpy as np, torch
from flygym_amd import _native
lib_prof = ROOT / "build" / "libnmf_prof.so"      # (a diagnostic build: never beside the product library)
lib_prof.parent.mkdir(exist_ok=True)
if "--build" in sys.argv or not lib_prof.exists():
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-mllvm", "-amdgpu-sched-strategy=iterative-ilp", *_native.MATH_FLAGS,
                    "-mllvm", "-amdgpu-atomic-optimizer-strategy=None", "-fPIC", "-shared", "-DNMF_STAGE_PROFILE", *[a for a in sys.argv if a.startswith("-D")],
                    f"-I{ROOT/'include'}", f"-I{ROOT/'flygym_amd/csrc'}", str(ROOT/"flygym_amd/csrc/nmf_capi.hip"), "-o", str(lib_prof)], check=True)
    if "--build" in sys.argv: sys.exit(0)
_native.LIB_PATH = lib_prof
from flygym_amd import HIPSimulation, make_model
from flygym_amd.compose import ActuatorType
from flygym_amd.replay import ReplayTargetData
n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 4096
preset = next((a.split('=')[1] for a in sys.argv if a.startswith('--joint-preset=')), 'legs_only')
fly, world, _ = make_model(joints_preset=preset)
terrain = next((a.split('=')[1] for a in sys.argv if a.startswith('--terrain=')), 'flat')
if terrain != 'flat':
    import flygym_amd.compose as C
    from flygym_amd.utils.math import Rotation3D
    world = {"gapped": C.GappedTerrainWorld, "blocks": C.BlocksTerrainWorld, "mixed": C.MixedTerrainWorld}[terrain]()
    world.add_fly(fly, (0, 0, 0.8), Rotation3D("quat", (1, 0, 0, 0)))
sim = HIPSimulation(world, n_worlds=n, device=0)
L = _native.lib()
L.nmf_debug_stage_cycles.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
order = fly.get_actuated_jointdofs_order(ActuatorType.POSITION)
if preset == "all_possible":      # the walking clip has no angles for this skeleton's extra axes: the CPG holds them at zero
    from flygym_amd.controllers import TripodCPG
    table = TripodCPG(order, 1e-4).targets(n, 1000, device=sim.device)
else:
    table = torch.as_tensor(ReplayTargetData(1e-4, order).make_target_angles_all_worlds(n, 1000), device=sim.device)
ids = sim._ids_by_fly[fly.name]["actuators"][ActuatorType.POSITION]
sim.set_leg_adhesion_states(fly.name, np.ones((n, 6), dtype=np.float32))
sim.step(500); torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 48)()
L.nmf_debug_stage_cycles(buf, 48, 1)
steps = next((int(a.split('=')[1]) for a in sys.argv if a.startswith('--steps=')), 500)
sim.step_replay(table, ids, 0, steps); torch.cuda.synchronize()
L.nmf_debug_stage_cycles(buf, 48, 1)
names = ["ctrl load", "kinematics", "inertia", "collision", "contact params", "velocity+bias", "actuation+project",
         "ABA smooth", "solver start (primal: first gradient; dual: j0, je, e.M.e)", "primal: test/exit; dual: responses + A", "primal: ABA(H); dual: elimination", "newton: jv, g1, g2 / row sums", "newton: linesearch",
         "newton: move", "final forces (dual: qacc expansion + wrenches)", "integrate (ABA Euler)", "write outputs", "sensors",
         "(all ABA) rest up", "(all ABA) legs + root", "(all ABA) rest down",
         "(collision) parameters + cull + capsules", "(collision) hull scans", "(collision) slots + contact ranges", "(collision) hulls scanned per step", "(collision) scans without a contact per step", "(collision) their summed dmin [nm]", "(collision) scans with contacts per step",
         "(hull) setup", "(hull) first scan + argmin", "(hull) patch scans", "(hull) contact output", "(hull) one-cell hulls per step", "(hull) vertices scanned per step", "(hull) patch scans over the candidate list per step", "-",
         "(expansion) clear the hinge sums", "(expansion) direction forces + LDS adds per hinge", "(expansion) six wave sums for the root", "(expansion) root", "(expansion) legs, root to leaf", "-", "-", "(contact space) direction responses, leaf to root", "(contact space) Gram matrix of the directions"]
cyc = np.array(list(buf)[:len(names)], dtype=np.float64) / steps
tot = cyc[:18].sum()
print(f"n_worlds {n}: wave-0 cycles per step = {tot:.0f}  (iters {sim.field('stats')[:,1].mean().item():.2f}, contacts {sim.field('stats')[:,0].mean().item():.2f})")
for nm, c in zip(names, cyc):
    print(f"  {nm:62s} {c:9.2f}  {100*c/tot:5.1f}%")
