"""Two MI355X over RCCL (skipped on a 1-GPU box — the driver's round-end box has one; kept for the day it has eight): the same
population sharded over two ranks steps bit for bit like the single-GPU run (one kernel per skeleton whatever the batch size,
csrc/nmf_dual.h), the observation all-gather delivers every world's row on every rank, and ``bench.py --gpus 2 --scaling
strong`` reports two RCCL ranks.  The world_size-2 logic itself is covered on CPU over gloo (tests/test_sharding_gloo.py)."""
import json
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]
N, STEPS = 1024, 150


def _need_two():
    import torch

    if not torch.cuda.is_available() or torch.cuda.device_count() < 2:
        pytest.skip("needs two MI355X")
    return torch


def _rank(rank, world_size, port, out):
    sys.path.insert(0, str(ROOT))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch
    import torch.distributed as dist

    from flygym_amd import HIPSimulation, make_model
    from flygym_amd.controllers import TripodCPG
    from flygym_amd.sharding import ObsGather, shard_range

    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world_size, device_id=torch.device("cuda", rank))      # RCCL
    fly, world, _ = make_model()
    first, last = shard_range(N, rank, world_size)
    n_local = last - first
    sim = HIPSimulation(world, n_worlds=n_local, device=rank)
    table = TripodCPG(fly.get_actuated_jointdofs_order("position"), 1e-4).targets(n_local, 1250, device=sim.device, first_world=first, total_worlds=N)
    ids = sim.replay_ids(fly.name)
    sim.set_leg_adhesion_states(fly.name, np.ones((n_local, 6), dtype=np.float32))
    sim.step(500)
    gather = ObsGather(n_local, sim.model.nv - 6, 42, sim.device, total_worlds=N, packer=sim.pack_observations)
    for k in range(STEPS // 50):
        sim.step_replay(table, ids, 50 * k, 50)
        gather.tick(sim.field("qpos"), sim.field("qvel"), sim.field("actuator_force"), sim.field("sensordata"))
    full = gather.wait()[gather.rows()].clone()
    gather.drain()
    torch.cuda.synchronize()
    np.savez(f"{out}.{rank}.npz", obs=full.cpu().numpy(), qpos=sim.field("qpos").cpu().numpy(), first=first)
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_over_rccl_step_like_one_gpu(tmp_path):
    torch = _need_two()
    import torch.multiprocessing as mp

    from flygym_amd import HIPSimulation, make_model
    from flygym_amd.controllers import TripodCPG

    out = str(tmp_path / "rank")
    mp.spawn(_rank, args=(2, 29700 + os.getpid() % 1500, out), nprocs=2, join=True)
    fly, world, _ = make_model()
    sim = HIPSimulation(world, n_worlds=N, device=0)
    table = TripodCPG(fly.get_actuated_jointdofs_order("position"), 1e-4).targets(N, 1250, device=sim.device)
    ids = sim.replay_ids(fly.name)
    sim.set_leg_adhesion_states(fly.name, np.ones((N, 6), dtype=np.float32))
    sim.step(500)
    for k in range(STEPS // 50):
        sim.step_replay(table, ids, 50 * k, 50)
    row = torch.empty((N, 270), device=sim.device)
    sim.pack_observations(row)
    want_obs, want_q = row.cpu().numpy(), sim.field("qpos").cpu().numpy()
    r0, r1 = np.load(out + ".0.npz"), np.load(out + ".1.npz")
    assert np.array_equal(r0["obs"], r1["obs"])                                   # every rank holds every world's row
    assert np.array_equal(r0["obs"], want_obs)                                    # ... bit for bit the single-GPU run's
    assert np.array_equal(np.concatenate([r0["qpos"], r1["qpos"]]), want_q)      # the shards' states too


def test_bench_two_gpus_strong_scaling_reports_two_rccl_ranks():
    _need_two()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    args = ["--steps", "100", "--warmup", "50", "--no-cpu-baseline", "--no-live-counters", "--no-other-configs"]
    two = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--scaling", "strong", *args], capture_output=True, text=True, env=env, timeout=900, cwd=ROOT)
    lines = [ln for ln in two.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1, two.stdout[-2000:] + two.stderr[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["rccl_ranks"] == 2 and d["valid"] and d["scaling"] == "strong"
    assert d["config"]["total_worlds"] == 4096 and sorted(d["config"]["shard_sizes"]) == [2048, 2048]
