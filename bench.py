"""Headline benchmark: env-steps/s of the batched NeuroMechFly stepping path on N x MI355X.

Protocol = the reference's GPU benchmark (``src/flygym_demo/benchmark/time_gpu_simulation.py:108-198``,
driver ``scripts/dev/run_gpu_benchmark.py:10-32``): benchmark model (``make_model`` defaults: LEGS_ONLY
skeleton nq 73 / nv 72, 42 position actuators kp 50, 6 adhesion actuators, flat ground, 55 geom-plane
pairs, mesh-hull collision geometry), adhesion on for all legs, the UNCONDITIONAL untimed settle of the
reference (``sim.warmup()`` = 500 steps at the neutral targets, ``:129-131``) — and, for the CPG
workload, one more full 12 Hz gait cycle under the control table so that the timed region is steady
walking — then W further untimed warm-up steps, then K timed steps, no rendering;
``steps_per_second = K * n_worlds / wall``.

    python bench.py                      # 1 GPU, 4096 worlds, 1000 timed steps
    python bench.py --gpus 8             # spawns 8 ranks itself (one process per GPU, RCCL over xGMI)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        --master-port 29500 bench.py --gpus 8 --steps 1000 --warmup 500       # the same, launched from outside

A short timed region (K < 500: the driver's ``--steps 20``) is one launch of a few milliseconds, so ``repeats`` of
them run back to back (the gait continuing) inside one barrier + synchronize bracket and the mean region time is
reported; ``config.repeats`` says how many.  The line is marked
``"valid": false`` and the exit code is 3 if the timed region was not contact-rich stepping (mean contacts < 1
or no solver iterations: flies in free fall), if the state went non-finite, or if contacts overflowed.

One process per GPU; worlds are independent, so each rank steps its own shard (``--scaling weak``: 4096 per
GPU; ``--scaling strong``: 4096 in total) and the only inter-GPU traffic is an RCCL all-gather of the
observation block once per control tick.  Rank 0 prints ONE JSON line.
"""

from __future__ import annotations

import argparse
import json
import math
import os
import socket
import subprocess
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
for p in (ROOT, ROOT / "oracle"):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))

# algorithmic HBM bytes per env-step (f32): read qpos 73 + qvel 72 + ctrl 48 +
# qacc_warmstart 72, write qpos 73 + qvel 72 + qacc_warmstart 72  = 482 floats.
BYTES_PER_ENV_STEP = 1928
HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec
# eye renderer: arithmetic of one ray of the common case: theta^2 3 flop, the two lens polynomials 15 fma, the
# direction 6 fma + 3 mul, the ground hit and checker parity ~14, the byte selection / run sums are integer work
EYE_FLOP_PER_RAY = 3 + 30 + 15 + 14
VALU_PEAK_TFLOPS = 157.3       # same guide: FP32 vector peak (256 CUs x 4 SIMD-32 x 2 flop x 2.4 GHz)
OBS_DIM = 66 + 66 + 42 + 96    # joint angles, joint velocities, actuator forces, contact sensors
SETTLE_NEUTRAL_S = 0.05        # reference Simulation.warmup(duration_s=0.05)
GAIT_HZ = 12.0


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=500,
                    help="untimed steps under the control table AFTER the unconditional settle")
    ap.add_argument("--repeats", type=int, default=0,
                    help="how often the K-step timed region is repeated (0 = auto: about 1000 timed steps in total)")
    ap.add_argument("--worlds-per-gpu", type=int, default=4096)
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak: --worlds-per-gpu on every GPU; strong: --worlds-per-gpu worlds in TOTAL, sharded over the GPUs")
    ap.add_argument("--shard-policy", choices=["spread", "fill"], default="spread",
                    help="--scaling strong only.  spread: every GPU gets a share (highest aggregate rate: a step takes 61 us for "
                         "128 worlds and 72 us for 2048); fill: use only ceil(total / resident worlds per GPU) GPUs — the same "
                         "rate to within ~10 %% from far fewer GPUs; the other ranks idle (flygym_amd.sharding.shard_plan)")
    ap.add_argument("--steps-per-launch", type=int, default=50,
                    help="physics steps fused into one kernel launch (= one control tick)")
    ap.add_argument("--workload", choices=["cpg", "replay"], default="cpg",
                    help="cpg: BASELINE config 2 (position-actuated tripod CPG); replay: the reference benchmark's "
                         "kinematic replay of the Spotlight clip (world w <- partition w %% 20)")
    ap.add_argument("--terrain", choices=["flat", "gapped", "blocks", "mixed"], default="flat",
                    help="flat = BASELINE config 2; gapped/blocks = config 4; mixed = config 5 (build-defined height maps)")
    ap.add_argument("--cpg-adhesion", type=float, default=0.0, metavar="ON",
                    help="drive leg adhesion from the CPG: control ON in stance, 1 (the reference's minimum) in swing (config 5)")
    ap.add_argument("--joint-preset", choices=["legs_only", "legs_active_only", "all_biological", "all_possible"], default="legs_only",
                    help="skeleton: the benchmark's LEGS_ONLY (default), the 48-dof LEGS_ACTIVE_ONLY, or the full-body ALL_BIOLOGICAL / "
                         "ALL_POSSIBLE (hybrid kernels)")
    ap.add_argument("--odor", action="store_true", help="evaluate the four odor sensors every control tick (config 5)")
    ap.add_argument("--vision", choices=["off", "resample", "render"], default="off",
                    help="BASELINE config 3: per vision tick (every --vision-every physics steps = one launch) both 512 x 450 eye "
                         "frames of every fly become 2 x 721 ommatidia readings.  resample: synthetic raw frames resident in HBM "
                         "(seeded noise over a checker floor) through the retina kernel — the HBM-bound kernel of the path; "
                         "render: the eye views are ray-cast on the GPU, fused with the resample (no raw frames in HBM)")
    ap.add_argument("--vision-every", type=int, default=20, help="physics steps per vision tick (20 = 500 Hz)")
    ap.add_argument("--eye-rays", type=int, choices=[0, 16], default=0,
                    help="--vision render: rays per ommatidium; 0 = every pixel of the raw frame inside the lattice (readings = "
                         "resampling the rendered frame, bit for bit), 16 = the sampled mode (an approximation, 15 x fewer rays)")
    ap.add_argument("--simplify-geom", action="store_true", help="all-capsule collision geometry variant")
    ap.add_argument("--obs-every", type=int, default=0, metavar="K",
                    help="record the 270-float observation block (joint angles, velocities, actuator forces, contact sensors) of every "
                         "K-th physics step inside the fused launch (nmf_step_record): the reference's loops read them after every step. "
                         "The roofline then prices 3008 B per recorded env-step (SURVEY 8(d)); 0 = off (outputs on a launch's last step only)")
    ap.add_argument("--cpu-flavour", action="store_true",
                    help="step the engine of the reference's CPU class (flygym_amd.Simulation: Newton + option/noslip_iterations = 5, "
                         "mujoco_globals.yaml:15) instead of the batched class's, which strips the pass (BASELINE config 1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-live-counters", action="store_true",
                    help="do not re-run the workload under rocprofv3 --pmc after the timed region (N = 1): roofline.traffic / "
                         ".issue then come from the committed passes under profiles/ and say so")
    ap.add_argument("--cpu-steps", type=int, default=200000)
    ap.add_argument("--no-other-configs", action="store_true",
                    help="do not append the short runs of BASELINE configs 1, 3, 4 and 5 (`other_configs`, N = 1 only)")
    return ap.parse_args(argv)


def host_cores(cap=64):
    """CPUs this process can actually use: affinity mask and cgroup CPU quota (the GPU box shows 256 logical CPUs
    but the container is limited to a fraction of them), capped to keep the baseline sample bounded."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except (OSError, ValueError):
            pass
    return max(1, min(n, cap))


def cpu_baseline(model_blob, table_rows, act_ids, settle_steps, n_steps):
    """The C oracle (a port of the pipeline, NOT real MuJoCo) on the host cores, same control tables: one world on
    one core, then one world per core on all cores (independent oracle instances on Python threads; the C call
    releases the GIL)."""
    import threading

    import oracle as orc

    orc.build()

    def make():
        o = orc.Oracle(model_blob, "f64")
        o.ctrl[42:] = 1.0
        o.step(settle_steps)
        return o

    o = make()
    t0 = time.perf_counter()
    o.step_replay(table_rows[0], act_ids, 0, n_steps)
    res = [time.perf_counter() - t0]
    single = n_steps / res[0]
    cores = max(1, min(host_cores(), len(table_rows)))
    budget_s, chunk = 10.0, 2000           # every thread steps in chunks until the time budget is spent
    gate = threading.Barrier(cores + 1)
    done = [0] * cores

    def worker(k):
        ok = make()               # settle outside the timed region
        gate.wait()
        end = time.perf_counter() + budget_s
        pos = 0
        while time.perf_counter() < end:
            ok.step_replay(table_rows[k], act_ids, pos, chunk)
            pos += chunk
        done[k] = pos

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(cores)]
    for th in threads:
        th.start()
    gate.wait()
    t0 = time.perf_counter()
    for th in threads:
        th.join()
    wall = time.perf_counter() - t0
    total = sum(done)
    return {
        "value": total / wall, "unit": "env-steps/s", "cores": cores, "kind": "port",
        "single_core_value": single,
        "sample": f"float64 C oracle (oracle/nmf_oracle.c) on the same control tables after {settle_steps} settle steps: "
                  f"{cores} worlds x ~{total // cores} steps on {cores} threads ({os.cpu_count()} logical CPUs visible, "
                  f"{cores} usable by this process) in {wall:.1f} s; "
                  f"single core: 1 world x {n_steps} steps in {res[0]:.1f} s",
    }


def algorithmic_flops(nv, nb, ncon, iters):
    """Useful floating-point operations of one fly-step (DESIGN.md §3 "compute roofline"): the arithmetic of the
    articulated-body formulation itself, counted per stage for this model's sizes and the measured mean contact count
    and Newton iteration count — no shadow lanes, no address arithmetic, no reductions' redundant copies."""
    nh = nv - 6
    kin = nh * 45 + nb * 75 + nv * 21            # joint quaternions + relative transforms, chain products, motion subspaces
    inertia = nb * 140                           # R I R^T, parallel-axis shift
    collision = 55 * 40 + ncon * 60 + 6 * 400 * 8    # culls, ~6 near hulls of ~400 vertices (distance scan), contact frames
    bias = nv * 40 + nb * 130                    # velocities, velocity-product accelerations, I a + v x* I v
    aba = nh * 150 + 6 * 60 + nb * 36 + nv * 14  # per hinge: U = IA s, D, rank-1 downdate, bias; root; inertia rows; back-substitution
    newton = nv * 30 + nb * 70 + ncon * 160      # twists, I T, energy, row velocities, line search, merged gradient sweep
    solves = 2.0 + iters                         # smooth + Euler + one per Newton iteration
    return kin + inertia + collision + bias + solves * aba + iters * (newton + ncon * 50) + nv * 20


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_spawn(args):
    """``python bench.py --gpus N`` without an external launcher: re-run this script as N ranks under
    torch.distributed.run (one process per GPU) and pass rank 0's JSON line through."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), str(Path(__file__).resolve()), *sys.argv[1:]]
    return subprocess.run(cmd, env=env).returncode


def traffic_model(n_local, spl, args):
    """HBM bytes per launch cannot be counted from inside this process: they come from the committed rocprofv3 --pmc
    passes of this command (profiles/hbm_traffic.json, written by scripts/summarize_profile.py).  A persistent launch
    moves state + outputs once and one control-table row per step, so the profile stores both terms per world
    (fitted from two launch lengths) and any steps_per_launch scales from them."""
    tfile = ROOT / "profiles" / "hbm_traffic.json"
    if not tfile.exists() or args.terrain != "flat" or args.odor or args.cpg_adhesion or args.joint_preset != "legs_only":
        return None, None
    rec = json.loads(tfile.read_text())
    issue = rec.get("issue") or None
    if "per_world_per_launch_bytes" in rec:
        return n_local * (rec["per_world_per_launch_bytes"] + rec["per_world_per_step_bytes"] * spl), issue
    if (rec.get("worlds_per_gpu"), rec.get("steps_per_launch")) == (n_local, spl):
        return rec["traffic_bytes_per_launch"], issue
    return None, issue


LIVE_PASSES = (
    ("fetch", ["FETCH_SIZE"]), ("write", ["WRITE_SIZE"]),
    ("sq1", ["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_ANY"]),
    ("sq2", ["SQ_THREAD_CYCLES_VALU", "SQ_ACTIVE_INST_VALU", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"]),
)


def live_counters(spl, kernel, launches=6, budget_s=240.0, passes=None):
    """HBM traffic and SQ issue counters of THIS command measured by THIS run (N = 1): after the timed region, rank 0
    re-runs the same workload for a few launches under `rocprofv3 --kernel-trace --pmc ...` — FETCH_SIZE, WRITE_SIZE and
    two SQ groups, each in its own pass as MI355X_MICROARCH.md prescribes — and reads the counter CSVs (mean over the
    last launches of `kernel`).  Returns {counter: value per launch} or None when rocprofv3 is missing or a pass fails;
    the caller then falls back to the committed passes of the builder (profiles/hbm_traffic.json) and says so."""
    import csv
    import shutil
    import tempfile

    rocprof = shutil.which("rocprofv3")
    if rocprof is None:
        return None, "rocprofv3 not on PATH"
    if any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ):
        return None, "already running under a profiler"
    drop = {"--steps": 1, "--warmup": 1, "--repeats": 1, "--gpus": 1, "--steps-per-launch": 1, "--cpu-steps": 1, "--no-cpu-baseline": 0}
    argv, skip = [], 0
    for a in sys.argv[1:]:
        if skip:
            skip -= 1
            continue
        key = a.split("=", 1)[0]
        if key in drop:
            skip = drop[key] if "=" not in a else 0
            continue
        argv.append(a)
    inner = [sys.executable, str(Path(__file__).resolve()), *argv, "--no-cpu-baseline", "--no-live-counters", "--no-other-configs", "--steps-per-launch", str(spl),
             "--steps", str(spl * launches), "--warmup", "0", "--repeats", "1"]
    env = dict(os.environ, TMPDIR="/tmp")
    t0 = time.perf_counter()
    acc = {}
    with tempfile.TemporaryDirectory(dir="/tmp", prefix="nmf_live_") as tmp:
        for tag, ctrs in (passes or LIVE_PASSES):
            left = budget_s - (time.perf_counter() - t0)
            if left < 20:
                return None, "time budget of the live counter passes used up"
            cmd = [rocprof, "--kernel-trace", "--pmc", *ctrs, "--output-format", "csv", "-d", f"{tmp}/{tag}", "-o", "live", "--", *inner]
            try:
                r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=left)
            except subprocess.TimeoutExpired:
                return None, f"rocprofv3 pass {tag} timed out"
            files = sorted(Path(tmp, tag).rglob("*counter_collection.csv"))
            if r.returncode != 0 or not files:
                return None, f"rocprofv3 pass {tag} failed (rc {r.returncode})"
            by_disp = {}
            for row in csv.DictReader(open(files[0])):
                if kernel in row["Kernel_Name"]:
                    by_disp.setdefault(int(row["Dispatch_Id"]), {})[row["Counter_Name"]] = float(row["Counter_Value"])
            disp = [by_disp[k] for k in sorted(by_disp)][-max(1, launches - 2):]      # the inner run's timed launches
            if not disp:
                return None, f"no dispatch of {kernel} in pass {tag}"
            for c in ctrs:
                vals = [d[c] for d in disp if c in d]
                if not vals:
                    return None, f"counter {c} missing in pass {tag}"
                acc[c] = float(np.mean(vals))
    return acc, f"measured by this run: rocprofv3 --kernel-trace --pmc passes ({', '.join(t for t, _ in (passes or LIVE_PASSES))}) over {launches} launches of the same workload, {time.perf_counter() - t0:.0f} s"


def idle_rank(args, torch, dist, device, total_worlds, shard_sizes, rank, world_size, n_ticks):
    """A rank that holds no worlds (--shard-policy fill): it takes part in every collective of the job — the per-tick
    observation all-gather (with an empty block), the barriers of the timing bracket, the result reductions — in the
    order the working ranks issue them, and nothing else."""
    from flygym_amd.sharding import ObsGather

    nj = PRESET_NV[args.joint_preset] - 6
    gather = ObsGather(0, nj, 42, device, total_worlds=total_worlds, shard_sizes=shard_sizes)
    empty = [torch.zeros((0, w), dtype=torch.float32, device=device) for w in (7 + nj, 6 + nj, 48, 96)]

    def fence():
        torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()

    gather.tick(*empty)
    fence(); fence()
    for _ in range(n_ticks - 1):
        gather.tick(*empty)
    gather.drain()
    fence()
    dist.all_reduce(torch.zeros(1, dtype=torch.float64, device=device), op=dist.ReduceOp.MAX)
    dist.all_reduce(torch.zeros(4, dtype=torch.float64, device=device), op=dist.ReduceOp.SUM)
    dist.all_reduce(torch.ones(1, dtype=torch.float64, device=device), op=dist.ReduceOp.MIN)
    probe = torch.zeros(world_size, dtype=torch.int32, device=device)
    dist.all_gather_into_tensor(probe, torch.tensor([rank + 1], dtype=torch.int32, device=device))
    dist.barrier()
    dist.destroy_process_group()


PRESET_NV = {"legs_only": 72, "legs_active_only": 48, "all_biological": 132, "all_possible": 210}


def run(args, primary=True):
    """One workload, settled, timed and gated.  primary: the command line's own workload (distributed set-up, live
    counters, CPU baseline); otherwise one of the short `other_configs` runs in the same process (N = 1)."""
    env_ws = os.environ.get("WORLD_SIZE") if primary else "1"

    import torch
    import torch.distributed as dist

    from flygym_amd import HIPSimulation, make_model
    from flygym_amd.compose import ActuatorType
    from flygym_amd.replay import ReplayTargetData
    from flygym_amd.sharding import ObsGather, resident_worlds, shard_plan

    world_size = int(env_ws or "1")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world_size
    torch.cuda.set_device(local_rank)
    # NMF_BENCH_FORCE_DIST=1 exercises the RCCL code path (process group, all-gather, barrier, max-reduce)
    # even with one rank, so it can be validated on a single-GPU box
    use_dist = primary and (world_size > 1 or bool(os.environ.get("NMF_BENCH_FORCE_DIST")))
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    resident = resident_worlds(PRESET_NV[args.joint_preset])
    if args.scaling == "strong":
        total_worlds = args.worlds_per_gpu
        if total_worlds < world_size and args.shard_policy == "spread":
            raise SystemExit("more ranks than worlds")
        shard_sizes = shard_plan(total_worlds, world_size, resident, args.shard_policy)
    else:
        total_worlds = args.worlds_per_gpu * world_size
        shard_sizes = [args.worlds_per_gpu] * world_size
    first_world = sum(shard_sizes[:rank])
    n_local = shard_sizes[rank]
    active_gpus = sum(1 for x in shard_sizes if x > 0)
    shard_note = None
    if max(shard_sizes) < resident and world_size > 1:
        # a launch below residency is pure latency: a step takes as long for 128 worlds as for 2048 (measured on BASELINE
        # config 5's workload: 61.2 us at 128 worlds per GPU, 66.2 at 512, 67.6 at 1024, 72.3 at 2048: scripts/gpu_config5_sweep.sh), so this job cannot scale with the GPU count
        shard_note = (f"{max(shard_sizes)} worlds per GPU are below the {resident} one MI355X steps at once: the step time no longer "
                      f"falls with the shard size, so {active_gpus} GPUs deliver about what {max(1, -(-total_worlds // resident))} would "
                      "(--shard-policy fill uses only those)")
        if rank == 0:
            print("bench.py: " + shard_note, file=sys.stderr)
    if args.vision != "off":
        args.steps_per_launch = args.vision_every      # one launch per vision tick
    spl = max(1, min(args.steps_per_launch, args.steps))
    if args.steps % spl:
        spl = next(d for d in range(spl, 0, -1) if args.steps % d == 0)
    n_launches = args.steps // spl
    repeats = args.repeats if args.repeats > 0 else (1 if args.steps >= 500 else min(100, max(2, math.ceil(1000 / args.steps))))

    if n_local == 0:         # --shard-policy fill left this rank without worlds: it only keeps the collectives company
        idle_rank(args, torch, dist, torch.device("cuda", local_rank), total_worlds, shard_sizes, rank, world_size,
                  1 + repeats * n_launches)
        return None, True, rank
    fly, world, _ = make_model(joints_preset=args.joint_preset, simplify_geom=args.simplify_geom)
    if args.terrain != "flat":
        import flygym_amd.compose as C
        from flygym_amd.utils.math import Rotation3D

        world = {"gapped": C.GappedTerrainWorld, "blocks": C.BlocksTerrainWorld, "mixed": C.MixedTerrainWorld}[args.terrain]()
        world.add_fly(fly, (0, 0, 0.8), Rotation3D("quat", (1, 0, 0, 0)))
    sim = HIPSimulation(world, n_worlds=n_local, device=local_rank, _cpu_flavour=bool(getattr(args, "cpu_flavour", False)))
    odor = None
    if args.odor:
        from flygym_amd.sensors import OdorSensors

        rng = np.random.default_rng(0)     # S = 3 sources, D = 2 dims, seeded within +-20 mm
        src = rng.uniform(-20, 20, (3, 3)); src[:, 2] = rng.uniform(0.5, 3.0, 3)
        odor = OdorSensors(sim, fly.name, src, rng.uniform(0.1, 1.0, (3, 2)))
    see, frames, vision_out = None, None, None
    if args.vision == "resample":
        from flygym_amd.sensors import RAW_IMG_HEIGHT as H, RAW_IMG_WIDTH as W, Retina

        retina = Retina()
        g = torch.Generator(device=sim.device); g.manual_seed(first_world)
        frames = torch.randint(0, 256, (n_local, 2, H, W, 3), dtype=torch.uint8, device=sim.device, generator=g)
        yy, xx = torch.meshgrid(torch.arange(H, device=sim.device), torch.arange(W, device=sim.device), indexing="ij")
        frames[:, :, ((yy // 32 + xx // 32) % 2 == 0) & (yy > H // 2)] //= 4          # darker checker floor
        del yy, xx
        see = lambda: retina.raw_image_to_hex_pxls(frames)
    elif args.vision == "render":
        from flygym_amd.vision import EyeRenderer, Scene

        eyes = EyeRenderer(sim, fly.name, Scene(spheres=[(8.0, 3.0, 1.5, 1.0)], sphere_rgb=[(0.05, 0.05, 0.05)]), rays_per_ommatidium=args.eye_rays)
        see = lambda: eyes.render()
        # rays cast per eye view: the 16-pixel chunks of the raw image that touch an ommatidium (the rest of the frame is never
        # rendered) — or, in the sampled mode, 16 per ommatidium
        EYE_RAYS_PER_VIEW = (eyes.retina.num_ommatidia * args.eye_rays if args.eye_rays else
                             int((eyes.retina.id_map.reshape(-1, 16) > 0).any(axis=1).sum()) * 16)
    order = fly.get_actuated_jointdofs_order(ActuatorType.POSITION)
    if args.workload == "replay":
        table_steps = 1000  # clip partitions of 1000 steps, as in the reference benchmark
        replay = ReplayTargetData(sim.timestep, order, device=sim.device)      # smoothed + resampled on the GPU
        table = replay.make_target_angles_all_worlds(n_local, table_steps, first_world=first_world)
    else:
        from flygym_amd.controllers import TripodCPG

        table_steps = 2500  # three 12 Hz gait cycles: the table wraps around seamlessly
        cpg = TripodCPG(order, sim.timestep)
        adhesion = (cpg.stance_bins(sim.model, fly), args.cpg_adhesion, 1.0) if args.cpg_adhesion > 0 else None
        table = cpg.targets(n_local, table_steps, device=sim.device, first_world=first_world,
                            total_worlds=total_worlds, adhesion=adhesion)   # built on the GPU: no multi-GB host arrays
    act_ids = sim.replay_ids(fly.name, with_adhesion=args.workload == "cpg" and args.cpg_adhesion > 0)

    # ---- the unconditional settle (never a CLI knob): reference warm-up, then one gait cycle under the CPG table
    sim.set_leg_adhesion_states(fly.name, np.ones((n_local, 6), dtype=np.float32))
    settle_neutral = int(SETTLE_NEUTRAL_S / sim.timestep)
    sim.step(settle_neutral)
    cursor = 0                                   # position in the control table; the gait continues across all phases
    settle_gait = 0
    if args.workload == "cpg" and not os.environ.get("NMF_BENCH_R1_PROTOCOL"):   # (diagnostic: round-1 protocol for kernel A/B)
        settle_gait = int(math.ceil(1.0 / (GAIT_HZ * sim.timestep) / 50.0)) * 50       # >= one full 12 Hz cycle
        for _ in range(settle_gait // 50):
            sim.step_replay(table, act_ids, cursor, 50)
            cursor += 50
    done = 0
    while done < args.warmup:                    # the driver's --warmup: further untimed steps, same launch shape
        n = min(spl, args.warmup - done)
        sim.step_replay(table, act_ids, cursor, n)
        cursor += n; done += n

    # observation gather: double-buffered so that the RCCL all-gather of tick k runs on RCCL's stream while the
    # stepping kernel of tick k + 1 already runs on the compute stream (the only exchange of the path)
    nj = sim.model.nv - 6                # joint angles, joint velocities, position-actuator forces, contact sensors
    gather = ObsGather(n_local, nj, 42, sim.device, total_worlds=total_worlds, shard_sizes=shard_sizes,
                       packer=sim.pack_observations) if use_dist else None
    assert args.joint_preset != "legs_only" or 2 * nj + 42 + 96 == OBS_DIM

    n_events = repeats * n_launches
    ev0 = [torch.cuda.Event(enable_timing=True) for _ in range(n_events + 1)]
    ev1 = [torch.cuda.Event(enable_timing=True) for _ in range(n_events + 1)]
    ev2 = [torch.cuda.Event(enable_timing=True) for _ in range(n_events + 1)] if see is not None else None

    obs_every = int(getattr(args, "obs_every", 0) or 0)
    if obs_every and (spl % obs_every or obs_every > spl):
        raise SystemExit("--obs-every must divide --steps-per-launch")
    # the observation ring of a launch: one buffer, reused (a consumer would read it before the next tick)
    ring = torch.empty((spl // obs_every, n_local, OBS_DIM if args.joint_preset == "legs_only" else 2 * nj + 42 + 96), dtype=torch.float32, device=sim.device) if obs_every else None

    def control_tick(start, k=n_events):
        # the kernel is launched on torch's current stream, so these events bracket exactly it
        ev0[k].record()
        if ring is None:
            sim.step_replay(table, act_ids, start, spl)
        else:
            sim.record_into(ring, table, act_ids, start, spl, obs_every)
        ev1[k].record()
        if see is not None:                # the vision tick: ev1 .. ev2 bracket exactly the retina / eye kernel
            nonlocal vision_out
            vision_out = see()
            ev2[k].record()
        if odor is not None:
            odor.get_odor_intensities()
        if gather is not None:
            gather.tick(sim.field("qpos"), sim.field("qvel"), sim.field("actuator_force"), sim.field("sensordata"))

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # untimed: one tick to settle allocator / RCCL channels
    control_tick(cursor); cursor += spl
    fence()
    sim.shader_clock_hz(reset=True)              # the clock probe counts the timed region's launches only
    sums0 = sim.field("stats_sum").clone()
    # ONE bracket (barrier + synchronize on both sides) around `repeats` back-to-back K-step regions: the mean region time
    # is (t1 - t0) / repeats.  The observation gather of a tick overlaps the next tick's stepping kernel across region
    # boundaries as it does inside a region; every gather is drained inside the bracket.
    fence()
    t0 = time.perf_counter()
    for r in range(repeats):
        for k in range(n_launches):
            control_tick(cursor, r * n_launches + k)
            cursor += spl
    if gather is not None:
        gather.drain()
    fence()
    total_elapsed = time.perf_counter() - t0
    sums1 = sim.field("stats_sum").clone()
    shader_clock_hz = sim.shader_clock_hz()
    elapsed_t = torch.tensor([total_elapsed], dtype=torch.float64, device=sim.device)
    if use_dist:
        dist.all_reduce(elapsed_t, op=dist.ReduceOp.MAX)      # the slowest rank
    total_elapsed = float(elapsed_t.item())
    elapsed = total_elapsed / repeats

    # what the timed region actually stepped: means over its launches (kernel-side running sums, NMF_STATS_SUM)
    dsum = (sums1 - sums0).double().sum(dim=0)
    flags = torch.tensor([float(torch.isfinite(sim.field("qpos")).all().item())], dtype=torch.float64, device=sim.device)
    if use_dist:
        dist.all_reduce(dsum, op=dist.ReduceOp.SUM)
        dist.all_reduce(flags, op=dist.ReduceOp.MIN)
    steps_seen = max(float(dsum[0].item()), 1.0)
    mean_contacts, mean_iters = float(dsum[1].item()) / steps_seen, float(dsum[2].item()) / steps_seen
    overflow_steps = int(dsum[3].item())
    # how the steps' constraint solves ended (NMF_STATS_SUM columns 4..14), per million env-steps of the timed region
    per_m = lambda k: float(dsum[k].item()) * 1e6 / steps_seen
    solver_exits = {"unit": "per million env-steps", "contact_space": per_m(4), "kkt_exact": per_m(5), "tie_rule": per_m(6),
                    "stalled_line_search": per_m(7), "cost_tests": per_m(8), "iteration_limit": per_m(9), "primal_loop": per_m(10),
                    "fallback_resolves": per_m(11), "big_eliminations": per_m(12), "noslip_skipped": per_m(13), "no_contact": per_m(14)}
    finite = bool(flags[0].item() > 0)
    rccl_ranks = None
    if use_dist:
        probe = torch.zeros(world_size, dtype=torch.int32, device=sim.device)
        dist.all_gather_into_tensor(probe, torch.tensor([rank + 1], dtype=torch.int32, device=sim.device))
        rccl_ranks = int((probe > 0).sum().item()) if dist.get_world_size() == world_size else 0
    valid = finite and mean_contacts >= 1.0 and mean_iters > 0.0 and overflow_steps == 0 \
        and abs(steps_seen - float(total_worlds) * args.steps * repeats) < 0.5

    out = None
    if rank == 0:
        # mean kernel duration over the launches of the timed region (HIP events on the launch stream)
        kernel_ms = [ev0[k].elapsed_time(ev1[k]) for k in range(n_events)]
        ms = float(np.mean(kernel_ms))
        value = total_worlds * args.steps / elapsed
        # SURVEY §8(d) formula on this model's sizes: read qpos + qvel + ctrl + warm start, write qpos + qvel + warm start
        bytes_per_env_step = 4 * (2 * sim.model.nq + 4 * sim.model.nv + sim.model.nu)
        assert args.joint_preset != "legs_only" or bytes_per_env_step == BYTES_PER_ENV_STEP
        if obs_every:      # + the observation block on the recorded steps (1928 + 1080 = 3008 B with obs on every step)
            bytes_per_env_step += 4.0 * ring.shape[2] / obs_every
            assert args.joint_preset != "legs_only" or obs_every != 1 or bytes_per_env_step == 3008
        achieved = bytes_per_env_step * n_local * spl / (ms * 1e-3) / 1e9
        traffic, issue = traffic_model(n_local, spl, args)
        traffic_source = "profiles/hbm_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command run by the builder, fitted per world and per step)" if traffic is not None else None
        issue_source = "profiles/hbm_traffic.json (rocprofv3 --pmc SQ_* passes of this command run by the builder)" if issue else None
        live, live_note = None, ("off (--no-live-counters)" if args.no_live_counters else "N > 1: committed passes" if world_size > 1
                                 else "vision run: the live passes go to the retina kernel (`roofline`), this block uses the committed ones")
        if primary and not args.no_live_counters and world_size == 1 and see is None:
            live, live_note = live_counters(spl, "nmf_step_kernel")
        if live:
            # MI355X_MICROARCH.md, HBM section: FETCH_SIZE / WRITE_SIZE count KiB; FETCH_SIZE under-reports wide reads by 2x on
            # gfx950 (dword-wide state loads here, so x2 is the upper bound); quad-cycle SQ counters
            traffic = 1024.0 * (2.0 * live["FETCH_SIZE"] + live["WRITE_SIZE"])
            traffic_source = issue_source = live_note
            env_steps = float(n_local * spl)
            base = issue or {}
            issue = {"valu_insts_per_env_step": live["SQ_INSTS_VALU"] / env_steps, "salu_insts_per_env_step": live["SQ_INSTS_SALU"] / env_steps,
                     "lds_insts_per_env_step": live["SQ_INSTS_LDS"] / env_steps, "wave_cycles_per_env_step": 4.0 * live["SQ_WAVE_CYCLES"] / env_steps,
                     "valu_active_per_wave": live["SQ_ACTIVE_INST_VALU"] / live["SQ_WAVE_CYCLES"], "wait_any_per_wave": live["SQ_WAIT_ANY"] / live["SQ_WAVE_CYCLES"],
                     # pipe cycles per wave64 VALU instruction: the microbenchmark's figure (profiles/valu_issue_microbench.json)
                     "valu_cycles_per_inst": base.get("valu_cycles_per_inst", 1.66),
                     "active_lane_fraction": live["SQ_THREAD_CYCLES_VALU"] / (64.0 * live["SQ_ACTIVE_INST_VALU"]) if live["SQ_ACTIVE_INST_VALU"] else None,
                     "lds_bank_conflict_fraction": live["SQ_LDS_BANK_CONFLICT"] / live["SQ_LDS_IDX_ACTIVE"] if live["SQ_LDS_IDX_ACTIVE"] else None}
        flops = algorithmic_flops(sim.model.nv, sim.model.nb, mean_contacts, mean_iters)
        kernel_rate = n_local * spl / (ms * 1e-3)           # env-steps/s of the kernel alone on this GPU
        compute = {
            "algorithmic_flop_per_env_step": flops,
            "useful_tflops": flops * kernel_rate / 1e12,
            "frac_of_f32_vector_peak": flops * kernel_rate / 1e12 / VALU_PEAK_TFLOPS,
            "peak_tflops": VALU_PEAK_TFLOPS,
        }
        compute["shader_clock_hz"] = shader_clock_hz       # measured inside the timed region's launches (nmf_shader_clock)
        if issue and issue.get("valu_insts_per_env_step") and issue.get("valu_cycles_per_inst") and shader_clock_hz > 0:
            # vector-pipe occupancy: instructions/s (the committed profile's count per env-step x this run's kernel rate) x
            # the pipe cycles a wave64 VALU instruction takes (scripts/valu_issue_microbench.hip) / (1024 SIMDs x the clock
            # THIS kernel ran at — not the throttled clock of the microbenchmark)
            compute["valu_pipe_busy"] = issue["valu_insts_per_env_step"] * kernel_rate * issue["valu_cycles_per_inst"] / (1024 * shader_clock_hz)
        out = {
            "metric": "env-steps/sec (whole node), 4096 flies per GPU, flat terrain" + (
                f", vision on (2 x 721-ommatidia retina per {args.vision_every} steps)" if args.vision != "off" else ""),
            "value": value, "unit": "env-steps/s", "n_gpus": world_size, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": args.scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic", "valid": valid,
            "config": {
                "workload": f"{total_worlds} flies ({n_local}/GPU), {args.terrain} terrain" + (" + odor sensors" if args.odor else "") + (f" + CPG-driven adhesion ({args.cpg_adhesion:g} in stance)" if args.cpg_adhesion > 0 else "") + ", " + (f"{args.joint_preset.upper()} fly (nq {sim.model.nq}, nv {sim.model.nv}, nu {sim.model.nu}), {sim.model.ng} geom-plane pairs ")
                            + ("(capsule geoms)" if args.simplify_geom else "(mesh convex hulls + capsule claws)")
                            + (", position-actuated tripod CPG gait (12 Hz, per-world phase offsets; BASELINE config 2)"
                               if args.workload == "cpg" else
                               ", position-actuated kinematic replay of the Spotlight tripod-walking clip "
                               "(reference benchmark protocol)") + ", adhesion on"
                            + ({"off": "", "resample": f"; vision on (BASELINE config 3): every {args.vision_every} steps both 512 x 450 raw eye "
                                "frames of every fly (synthetic: seeded noise over a checker floor, resident in HBM) -> 2 x 721 x 2 ommatidia readings",
                                "render": f"; vision on (BASELINE config 3): every {args.vision_every} steps both eye views of every fly are ray-cast "
                                "(checker ground, sky, one sphere, the fly's own body) and resampled to 2 x 721 x 2 ommatidia readings in one kernel"}[args.vision]),
                "control": args.workload,
                # which class's engine: "batched" = GPUSimulation's (noslip stripped, warp/simulation.py:427-448), "cpu" = Simulation's
                # (Newton + 5 noslip sweeps per step, simulation.py:74-76 under mujoco_globals.yaml:15)
                "engine_flavour": "cpu" if getattr(args, "cpu_flavour", False) else "batched",
                "noslip_iterations": sim.batch_info()["noslip_iterations"],
                "worlds_per_gpu": n_local, "total_worlds": total_worlds, "steps_per_launch": spl,
                # pure outputs (segment poses, contact sensors, actuator forces) are computed on a launch's last step only: a
                # caller of nmf_step(n) cannot observe the intermediate ones (the reference's captured loop computes them every step)
                "outputs_every_steps": spl,
                # the observation block (joint angles, velocities, actuator forces, contact sensors: what the reference's loops read
                # after every step) is recorded inside the launch on every obs_every-th step (nmf_step_record); 0 = not recorded
                "obs_every_steps": obs_every,
                "batch_info": sim.batch_info(),
                "shard_sizes": shard_sizes, "active_gpus": active_gpus, "resident_worlds_per_gpu": resident, "shard_note": shard_note,
                "settle_steps": {"neutral": settle_neutral, "gait": settle_gait, "warmup": args.warmup},
                "repeats": repeats, "timed_steps_total": args.steps * repeats,
                "elapsed_s": {"total": total_elapsed, "per_region": elapsed},
                "kernel_ms_per_launch": {"mean": ms, "min": float(min(kernel_ms)), "max": float(max(kernel_ms))},
                "timestep": sim.timestep, "realtime_factor": value * sim.timestep,
                "parallelism": f"env-shard x{world_size}" + (", RCCL all-gather of obs per control tick" if use_dist else ""),
                "rccl_ranks": rccl_ranks,
                "state_finite": finite, "contact_overflow_steps": overflow_steps,
                "mean_contacts": mean_contacts, "mean_newton_iters": mean_iters,
                "solver_exits": solver_exits,
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                # traffic / issue: at N = 1 measured by THIS run (live_counters: the workload re-run under rocprofv3 --pmc after
                # the timed region); otherwise the committed passes of the same command under profiles/, scaled to this
                # launch length — `counters` / `*_source` say which
                "traffic_source": traffic_source, "issue_source": issue_source,
                "counters": "live" if live else "replayed", "counters_note": live_note,
                "kernel": {"legs_only": "nmf_step_kernel<HybridTopo<0,0,6,3,2,1,1,1,1,1,1>, false>",
                           "legs_active_only": "nmf_step_kernel<HybridTopo<0,0,6,3,2,1,1>, false>",
                           "all_biological": "nmf_step_kernel<HybridTopo<20,60,6,3,2,1,1,1,1,1,1>, false>",
                           "all_possible": "nmf_step_kernel<HybridTopo<20,60,6,3,3,3,3,3,3,3,3>, false>"}[args.joint_preset],
                "kernel_ms_per_launch": ms,
                "algorithmic_bytes_per_env_step": bytes_per_env_step,
                # instruction-issue side of the same kernel, from the SQ counters of the committed profile (profiles/*_summary.md)
                "issue": issue,
                "compute": compute,
                "note": "the step is bound by dependent-instruction latency / VALU issue, not by HBM (state crosses HBM once "
                        "per launch); see DESIGN.md for the instruction-side analysis",
            },
        }
        if see is not None:
            # BASELINE config 3: the retina kernel is the HBM-bound kernel of the path, so IT carries the `roofline` block
            # (2 x 512 x 450 x 3 B of raw frames in + 2 x 721 x 2 x 4 B of readings out per fly and tick);
            # the stepping kernel's block stays beside it as `roofline_physics`.
            vis_ms = float(np.mean([ev1[k].elapsed_time(ev2[k]) for k in range(n_events)]))
            out["roofline_physics"] = out["roofline"]
            frame_bytes = (int(frames[0, 0].numel()) if frames is not None else 0)
            out_bytes = int(vision_out[0, 0].numel()) * 4
            per_launch = 2 * n_local * (frame_bytes + out_bytes)
            vt = None
            vfile = ROOT / "profiles" / "vision_traffic.json"
            if vfile.exists() and args.vision == "resample":
                rec = json.loads(vfile.read_text())
                vt = rec.get("traffic_bytes_per_eye_frame", 0) * 2 * n_local or None
            vt_source = "profiles/vision_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command run by the builder)" if vt else None
            vlive = None
            if primary and not args.no_live_counters and world_size == 1 and args.vision == "resample":
                vlive, vnote = live_counters(spl, "nmf_retina_stream_kernel", passes=LIVE_PASSES[:2])
                if vlive:      # wide coalesced 16-byte reads: the x2 of the guide's gfx950 correction applies in full
                    vt, vt_source = 1024.0 * (2.0 * vlive["FETCH_SIZE"] + vlive["WRITE_SIZE"]), vnote
            ach = per_launch / (vis_ms * 1e-3) / 1e9
            peak, unit = HBM_PEAK_GBS, "GB/s"
            if args.vision == "render":
                # compute-bound: the arithmetic a ray of the common case needs (EYE_FLOP_PER_RAY) x the rays
                # cast, against the f32 vector peak
                n_rays = 2 * n_local * EYE_RAYS_PER_VIEW
                ach, peak, unit = n_rays * EYE_FLOP_PER_RAY / (vis_ms * 1e-3) / 1e12, VALU_PEAK_TFLOPS, "TFLOP/s"
            out["roofline"] = {
                "bound": "hbm" if args.vision == "resample" else "valu", "achieved": ach, "peak": peak, "unit": unit,
                "frac": ach / peak, "traffic": vt,
                "traffic_source": vt_source, "counters": "live" if vlive else "replayed",
                "kernel": "nmf_retina_stream_kernel" if args.vision == "resample" else "nmf_eye_kernel",
                "kernel_ms_per_launch": vis_ms, "eye_frames_per_launch": 2 * n_local,
                "algorithmic_bytes_per_eye_frame": {"in": frame_bytes, "out": out_bytes},
                "algorithmic_bytes_per_launch": per_launch,
                "note": ("streams every raw frame once: HBM-bound by construction" if args.vision == "resample" else
                         "ray-casts the views instead of reading frames: compute-bound, the HBM figure is the readings written only"),
            }
            if args.vision == "render":
                out["roofline"]["rays_per_s"] = 2 * n_local * EYE_RAYS_PER_VIEW / (vis_ms * 1e-3)
                out["roofline"]["rays_per_view"] = EYE_RAYS_PER_VIEW
                out["roofline"]["algorithmic_flop_per_ray"] = EYE_FLOP_PER_RAY
            out["config"]["vision"] = {"mode": args.vision, "every_steps": args.vision_every, "kernel_ms_per_tick": vis_ms,
                                       "physics_kernel_ms_per_tick": ms}
            if args.vision == "render":
                out["config"]["vision"]["rays_per_ommatidium"] = args.eye_rays or "every pixel of the cell (pixel-exact)"
        if primary and not args.no_cpu_baseline and world_size == 1:   # reported at N=1 only
            n_rows = min(n_local, host_cores())
            rows = np.ascontiguousarray(table[:n_rows, :, :42].cpu().numpy())
            out["cpu_baseline"] = cpu_baseline(sim.model.to_blob(), rows, np.arange(42, dtype=np.int32), settle_neutral,
                                               args.cpu_steps)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    del sim
    torch.cuda.empty_cache()
    return out, valid, rank


# BASELINE configs other than the headline's (config 2), as short validity-gated runs appended to the line at N = 1
# (reference protocol: src/flygym_demo/benchmark/time_gpu_simulation.py:108-198; same settle, same gate, fewer timed steps)
OTHER_CONFIGS = (
    ("config 1: 1 fly, flat, kinematic replay, one step per launch, the CPU class's engine (flygym_amd.Simulation semantics: noslip 5 on)",
     dict(worlds_per_gpu=1, workload="replay", steps_per_launch=1, steps=200, cpu_flavour=True)),
    ("config 1's input on the batched class's engine (noslip stripped), one step per launch", dict(worlds_per_gpu=1, workload="replay", steps_per_launch=1, steps=200)),
    ("config 3: 4096 flies, vision every 20 steps, raw frames resampled", dict(vision="resample", steps=200)),
    ("config 3: 4096 flies, vision every 20 steps, eye views ray-cast", dict(vision="render", steps=200)),
    ("config 3: 4096 flies, vision every 20 steps, eye views ray-cast, 16 rays per ommatidium (sampled mode: an approximation)", dict(vision="render", eye_rays=16, steps=200)),
    ("config 2 with the observation block recorded on EVERY step inside the fused launches (3008 B/env-step, SURVEY 8(d))", dict(obs_every=1, steps=200)),
    ("config 2 stepped one launch per step (what a caller reading outputs between steps pays without the ring)", dict(steps_per_launch=1, steps=100)),
    ("config 4: 4096 flies per GPU, gapped terrain", dict(terrain="gapped", steps=200)),
    ("config 4: 4096 flies per GPU, blocks terrain", dict(terrain="blocks", steps=200)),
    ("config 5: 1024 flies, mixed terrain + odor sensors + gait-driven adhesion", dict(worlds_per_gpu=1024, terrain="mixed", odor=True, cpg_adhesion=20.0, steps=200)),
)


def other_configs(args):
    import argparse

    res = []
    t0 = time.perf_counter()
    for name, over in OTHER_CONFIGS:
        a = argparse.Namespace(**vars(args))
        a.gpus, a.scaling, a.warmup, a.repeats, a.no_cpu_baseline, a.no_live_counters = 1, "weak", 0, 0, True, True
        a.workload, a.terrain, a.odor, a.cpg_adhesion, a.vision, a.worlds_per_gpu, a.steps_per_launch = "cpg", "flat", False, 0.0, "off", 4096, 50
        a.joint_preset, a.simplify_geom, a.eye_rays, a.obs_every, a.cpu_flavour = "legs_only", False, 0, 0, False
        for k, v in over.items():
            setattr(a, k, v)
        try:
            o, ok, _ = run(a, primary=False)
            c = o["config"]
            rec = {"config": name, "value": o["value"], "unit": o["unit"], "valid": ok, "worlds": c["total_worlds"], "steps": a.steps,
                   "steps_per_launch": c["steps_per_launch"], "kernel_ms_per_launch": c["kernel_ms_per_launch"]["mean"],
                   "mean_contacts": c["mean_contacts"], "mean_newton_iters": c["mean_newton_iters"],
                   "engine_flavour": c["engine_flavour"], "noslip_iterations": c["noslip_iterations"]}
            if c.get("obs_every_steps"):
                rec["obs_every_steps"] = c["obs_every_steps"]
                rec["algorithmic_bytes_per_env_step"] = o["roofline"]["algorithmic_bytes_per_env_step"]
                rec["roofline_frac"] = o["roofline"]["frac"]
            if "vision" in c:
                rec["vision_kernel_ms_per_tick"] = c["vision"]["kernel_ms_per_tick"]
                rec["vision_kernel"] = o["roofline"]["kernel"]
                rec["vision_roofline_frac"] = o["roofline"]["frac"]
        except Exception as e:      # a broken side run must never take the headline down with it
            rec = {"config": name, "valid": False, "error": f"{type(e).__name__}: {e}"}
        res.append(rec)
    return res, time.perf_counter() - t0


def main():
    args = parse_args()
    if args.gpus > 1 and os.environ.get("WORLD_SIZE") is None:
        raise SystemExit(self_spawn(args))
    out, valid, rank = run(args, primary=True)
    if out is not None and out["n_gpus"] == 1 and not args.no_other_configs and not any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ):
        out["other_configs"], secs = other_configs(args)
        out["other_configs_seconds"] = secs
    if rank == 0:
        sys.stdout.flush()
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)   # RCCL's banner sits in C stdio buffers: push it out first
        except OSError:
            pass
        print(json.dumps(out), flush=True)   # the one JSON line, last thing on stdout
    if not valid:
        sys.exit(3)


if __name__ == "__main__":
    main()
