"""Headline benchmark: env-steps/s of the batched NeuroMechFly stepping path on N x MI355X.

Protocol = the reference's GPU benchmark (``src/flygym_demo/benchmark/time_gpu_simulation.py:108-198``,
driver ``scripts/dev/run_gpu_benchmark.py:10-32``): benchmark model (``make_model`` defaults: LEGS_ONLY
skeleton nq 73 / nv 72, 42 position actuators kp 50, 6 adhesion actuators, flat ground, 55 geom-plane
pairs, mesh-hull collision geometry), adhesion on for all legs, untimed warm-up, then K timed steps of
kinematic replay (world w replays clip partition w % 20), no rendering;
``steps_per_second = K * n_worlds / wall``.

    python bench.py                      # 1 GPU, 4096 worlds, 1000 timed steps after 500 warm-up steps
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        --master-port 29500 bench.py --gpus 8 --steps 1000 --warmup 500

One process per GPU; worlds are independent, so each rank steps its own 4096 (weak scaling) and the
only inter-GPU traffic is an RCCL all-gather of the observation block once per control tick.
Rank 0 prints ONE JSON line.
"""

from __future__ import annotations

import argparse
import json
import os
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
OBS_DIM = 66 + 66 + 42 + 96    # joint angles, joint velocities, actuator forces, contact sensors


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=500)
    ap.add_argument("--worlds-per-gpu", type=int, default=4096)
    ap.add_argument("--steps-per-launch", type=int, default=50,
                    help="physics steps fused into one kernel launch (= one control tick)")
    ap.add_argument("--workload", choices=["cpg", "replay"], default="cpg",
                    help="cpg: BASELINE config 2 (position-actuated tripod CPG); replay: the reference benchmark's "
                         "kinematic replay of the Spotlight clip (world w <- partition w %% 20)")
    ap.add_argument("--terrain", choices=["flat", "gapped", "blocks", "mixed"], default="flat",
                    help="flat = BASELINE config 2; gapped/blocks = config 4; mixed = config 5 (build-defined height maps)")
    ap.add_argument("--cpg-adhesion", type=float, default=0.0, metavar="ON",
                    help="drive leg adhesion from the CPG: control ON in stance, 1 (the reference's minimum) in swing (config 5)")
    ap.add_argument("--joint-preset", choices=["legs_only", "legs_active_only", "all_biological"], default="legs_only",
                    help="skeleton: the benchmark's LEGS_ONLY (default) or the full-body ALL_BIOLOGICAL (hybrid kernel)")
    ap.add_argument("--odor", action="store_true", help="evaluate the four odor sensors every control tick (config 5)")
    ap.add_argument("--simplify-geom", action="store_true", help="all-capsule collision geometry variant")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=200000)
    return ap.parse_args()


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


def cpu_baseline(model_blob, table_rows, act_ids, warmup, n_steps):
    """The C oracle (a port of the pipeline, NOT real MuJoCo) on the host cores, same control tables: one world on
    one core, then one world per core on all cores (independent oracle instances on Python threads; the C call
    releases the GIL)."""
    import threading

    import oracle as orc

    orc.build()

    def make():
        o = orc.Oracle(model_blob, "f64")
        o.ctrl[42:] = 1.0
        o.step(warmup)
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
        ok = make()               # warm-up outside the timed region
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
        "sample": f"float64 C oracle (oracle/nmf_oracle.c) on the same control tables after {warmup} warm-up steps: "
                  f"{cores} worlds x ~{total // cores} steps on {cores} threads ({os.cpu_count()} logical CPUs visible, "
                  f"{cores} usable by this process) in {wall:.1f} s; "
                  f"single core: 1 world x {n_steps} steps in {res[0]:.1f} s",
    }


def main():
    args = parse_args()
    import torch
    import torch.distributed as dist

    from flygym_amd import HIPSimulation, make_model
    from flygym_amd import _native
    from flygym_amd.compose import ActuatorType
    from flygym_amd.replay import ReplayTargetData

    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world_size:
        if world_size == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
        args.gpus = world_size
    torch.cuda.set_device(local_rank)
    # NMF_BENCH_FORCE_DIST=1 exercises the RCCL code path (process group, all-gather, barrier, max-reduce)
    # even with one rank, so it can be validated on a single-GPU box
    use_dist = world_size > 1 or bool(os.environ.get("NMF_BENCH_FORCE_DIST"))
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    n_local = args.worlds_per_gpu
    spl = max(1, min(args.steps_per_launch, args.steps))
    if args.steps % spl:
        spl = next(d for d in range(spl, 0, -1) if args.steps % d == 0)
    n_launches = args.steps // spl

    fly, world, _ = make_model(joints_preset=args.joint_preset, simplify_geom=args.simplify_geom)
    if args.terrain != "flat":
        import flygym_amd.compose as C
        from flygym_amd.utils.math import Rotation3D

        world = {"gapped": C.GappedTerrainWorld, "blocks": C.BlocksTerrainWorld, "mixed": C.MixedTerrainWorld}[args.terrain]()
        world.add_fly(fly, (0, 0, 0.8), Rotation3D("quat", (1, 0, 0, 0)))
    sim = HIPSimulation(world, n_worlds=n_local, device=local_rank)
    odor = None
    if args.odor:
        from flygym_amd.sensors import OdorSensors

        rng = np.random.default_rng(0)     # S = 3 sources, D = 2 dims, seeded within +-20 mm
        src = rng.uniform(-20, 20, (3, 3)); src[:, 2] = rng.uniform(0.5, 3.0, 3)
        odor = OdorSensors(sim, fly.name, src, rng.uniform(0.1, 1.0, (3, 2)))
    order = fly.get_actuated_jointdofs_order(ActuatorType.POSITION)
    replay = ReplayTargetData(sim.timestep, order)
    if args.workload == "replay":
        table_steps = 1000  # clip partitions of 1000 steps, as in the reference benchmark
        table_np = replay.make_target_angles_all_worlds(n_local, table_steps, first_world=rank * n_local)
    else:
        from flygym_amd.controllers import TripodCPG

        table_steps = 2500  # three 12 Hz gait cycles: the table wraps around seamlessly
        cpg = TripodCPG(order, sim.timestep)
        table_np = None
        adhesion = (cpg.stance_bins(sim.model, fly), args.cpg_adhesion, 1.0) if args.cpg_adhesion > 0 else None
        table = cpg.targets(n_local, table_steps, device=sim.device, first_world=rank * n_local,
                            total_worlds=n_local * world_size, adhesion=adhesion)   # built on the GPU: no multi-GB host arrays
    if table_np is not None:
        table = torch.as_tensor(table_np, device=sim.device)
    act_ids = sim.replay_ids(fly.name, with_adhesion=args.workload == "cpg" and args.cpg_adhesion > 0)
    maps = sim._ids_by_fly[fly.name]

    sim.set_leg_adhesion_states(fly.name, np.ones((n_local, 6), dtype=np.float32))
    if args.warmup > 0:
        sim.step(args.warmup)   # reference: sim.warmup() = 500 steps at the neutral targets

    # observation gather: double-buffered so that the RCCL all-gather of tick k runs on RCCL's stream while the
    # stepping kernel of tick k + 1 already runs on the compute stream (the only exchange of the path)
    nj = sim.model.nv - 6                # joint angles, joint velocities, position-actuator forces, contact sensors
    obs_dim = 2 * nj + 42 + 96
    assert args.joint_preset != "legs_only" or obs_dim == OBS_DIM
    obs_local = [torch.empty((n_local, obs_dim), dtype=torch.float32, device=sim.device) for _ in range(2)]
    obs_all = [torch.empty((world_size * n_local, obs_dim), dtype=torch.float32, device=sim.device) for _ in range(2)] if use_dist else None
    pending = [None, None]

    ev0 = [torch.cuda.Event(enable_timing=True) for _ in range(n_launches + 1)]
    ev1 = [torch.cuda.Event(enable_timing=True) for _ in range(n_launches + 1)]

    def control_tick(start, k=n_launches):
        # the kernel is launched on torch's current stream, so these events bracket exactly it
        ev0[k].record()
        sim.step_replay(table, act_ids, start, spl)
        ev1[k].record()
        if odor is not None:
            odor.get_odor_intensities()
        if use_dist:
            slot = k & 1
            if pending[slot] is not None:
                pending[slot].wait()          # the gather that last used this buffer pair (two ticks ago)
            ol = obs_local[slot]
            ol[:, 0:nj] = sim.field("qpos")[:, 7:]
            ol[:, nj:2 * nj] = sim.field("qvel")[:, 6:]
            ol[:, 2 * nj:2 * nj + 42] = sim.field("actuator_force")[:, :42]
            ol[:, 2 * nj + 42:] = sim.field("sensordata")
            pending[slot] = dist.all_gather_into_tensor(obs_all[slot], ol, async_op=True)

    # untimed: one tick to settle allocator / RCCL channels
    control_tick(0)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(n_launches):
        control_tick(spl * (k + 1), k)
    for w in pending:
        if w is not None:
            w.wait()                          # every gather is inside the timed region
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=sim.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    finite = bool(torch.isfinite(sim.field("qpos")).all().item())
    stats = sim.get_solver_stats()
    overflow = int(stats[:, 2].sum().item())

    if rank == 0:
        # mean kernel duration over the launches of the timed region (HIP events on the launch stream)
        ms = float(np.mean([ev0[k].elapsed_time(ev1[k]) for k in range(n_launches)]))
        total_worlds = n_local * world_size
        value = total_worlds * args.steps / elapsed
        # SURVEY §8(d) formula on this model's sizes: read qpos + qvel + ctrl + warm start, write qpos + qvel + warm start
        bytes_per_env_step = 4 * (2 * sim.model.nq + 4 * sim.model.nv + sim.model.nu)
        assert args.joint_preset != "legs_only" or bytes_per_env_step == BYTES_PER_ENV_STEP
        achieved = bytes_per_env_step * n_local * spl / (ms * 1e-3) / 1e9
        # HBM bytes per launch cannot be counted from inside this process: they come from the committed rocprofv3
        # --pmc passes of this very command (profiles/hbm_traffic.json, written by scripts/summarize_profile.py)
        traffic = issue = None
        tfile = ROOT / "profiles" / "hbm_traffic.json"
        if tfile.exists():
            rec = json.loads(tfile.read_text())
            if (rec.get("worlds_per_gpu"), rec.get("steps_per_launch"), rec.get("control")) == (n_local, spl, args.workload) \
                    and args.terrain == "flat" and not args.odor and not args.cpg_adhesion and args.joint_preset == "legs_only":
                traffic = rec["traffic_bytes_per_launch"]
                issue = rec.get("issue") or None
        out = {
            "metric": "env-steps/sec (whole node), 4096 flies per GPU, flat terrain",
            "value": value, "unit": "env-steps/s", "n_gpus": world_size, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": f"{n_local} flies/GPU, {args.terrain} terrain" + (" + odor sensors" if args.odor else "") + (f" + CPG-driven adhesion ({args.cpg_adhesion:g} in stance)" if args.cpg_adhesion > 0 else "") + ", " + (f"{args.joint_preset.upper()} fly (nq {sim.model.nq}, nv {sim.model.nv}, nu {sim.model.nu}), {sim.model.ng} geom-plane pairs ")
                            + ("(capsule geoms)" if args.simplify_geom else "(mesh convex hulls + capsule claws)")
                            + (", position-actuated tripod CPG gait (12 Hz, per-world phase offsets; BASELINE config 2)"
                               if args.workload == "cpg" else
                               ", position-actuated kinematic replay of the Spotlight tripod-walking clip "
                               "(reference benchmark protocol)") + ", adhesion on",
                "control": args.workload,
                "worlds_per_gpu": n_local, "total_worlds": total_worlds, "steps_per_launch": spl,
                "timestep": sim.timestep, "realtime_factor": value * sim.timestep,
                "parallelism": f"env-shard x{world_size}" + (", RCCL all-gather of obs per control tick" if use_dist else ""),
                "state_finite": finite, "contact_overflow_worlds": overflow,
                "mean_contacts": float(stats[:, 0].mean().item()), "mean_newton_iters": float(stats[:, 1].mean().item()),
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "kernel": {"legs_only": "nmf_step_kernel<HybridTopo<0,0,6,3,2,1,1,1,1,1,1>, false>",
                           "legs_active_only": "nmf_step_kernel<HybridTopo<0,0,6,3,2,1,1>, false>",
                           "all_biological": "nmf_step_kernel<HybridTopo<20,60,6,3,2,1,1,1,1,1,1>, false>",
                           "all_possible": "nmf_step_kernel<HybridTopo<20,60,6,3,3,3,3,3,3,3,3>, false>"}[args.joint_preset],
                "kernel_ms_per_launch": ms,
                "algorithmic_bytes_per_env_step": bytes_per_env_step,
                # instruction-issue side of the same kernel, from the SQ counters of the committed profile (profiles/*_summary.md)
                "issue": issue,
                "note": "the step is VALU-issue bound by construction (state crosses HBM once per launch); "
                        "see DESIGN.md for the instruction-side analysis",
            },
        }
        if not args.no_cpu_baseline and world_size == 1:   # reported at N=1 only
            n_rows = min(n_local, host_cores())
            rows = np.ascontiguousarray(table[:n_rows, :, :42].cpu().numpy())
            out["cpu_baseline"] = cpu_baseline(sim.model.to_blob(), rows, np.arange(42, dtype=np.int32), args.warmup,
                                               args.cpu_steps)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        sys.stdout.flush()
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)   # RCCL's banner sits in C stdio buffers: push it out first
        except OSError:
            pass
        print(json.dumps(out), flush=True)   # the one JSON line, last thing on stdout


if __name__ == "__main__":
    main()
