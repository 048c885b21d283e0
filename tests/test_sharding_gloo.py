"""Multi-GPU layout on CPU: 2 ranks over gloo exercise the sharding helpers and the observation
all-gather used by bench.py (on MI355X the same code runs over RCCL)."""

import os
import sys
from pathlib import Path

import numpy as np
import pytest

from flygym_amd.sharding import shard_range

ROOT = Path(__file__).resolve().parents[1]


def test_shard_range_partitions_exactly():
    for total, ws in [(4096 * 8, 8), (10, 3), (7, 7), (5, 8)]:
        spans = [shard_range(total, r, ws) for r in range(ws)]
        assert spans[0][0] == 0 and spans[-1][1] == total
        for (a0, a1), (b0, b1) in zip(spans[:-1], spans[1:]):
            assert a1 == b0 and 0 <= (a1 - a0) - (b1 - b0) <= 1
    with pytest.raises(ValueError):
        shard_range(10, 3, 3)


def _worker(rank, world_size, port, tmp):
    sys.path.insert(0, str(ROOT))
    import torch
    import torch.distributed as dist

    from flygym_amd.models import make_model
    from flygym_amd.replay import ReplayTargetData
    from flygym_amd.sharding import gather_observations, shard_range

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    fly = make_model()[0]
    order = fly.get_actuated_jointdofs_order("position")
    first, last = shard_range(12, rank, world_size)
    shard = ReplayTargetData(1e-4, order).make_target_angles_all_worlds(last - first, 1000, first_world=first)
    obs_local = torch.as_tensor(shard[:, 0, :]).clone()            # stands in for the obs block
    full = gather_observations(obs_local)
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)                       # bench.py's max-over-ranks timing
    if rank == 0:
        np.save(tmp, full.numpy())
        assert t.item() == world_size
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_gather(tmp_path):
    import torch.multiprocessing as mp

    from flygym_amd.models import make_model
    from flygym_amd.replay import ReplayTargetData

    out = tmp_path / "obs.npy"
    port = 29500 + os.getpid() % 2000
    mp.spawn(_worker, args=(2, port, str(out)), nprocs=2, join=True)
    fly = make_model()[0]
    order = fly.get_actuated_jointdofs_order("position")
    whole = ReplayTargetData(1e-4, order).make_target_angles_all_worlds(12, 1000)
    np.testing.assert_array_equal(np.load(out), whole[:, 0, :])


def _gather_worker(rank, world_size, port, tmp, total_worlds):
    """Drives bench.py's exchange — flygym_amd.sharding.ObsGather, double-buffered async all-gather — over gloo with
    stand-in engine fields that are overwritten between ticks, as the stepping kernel overwrites them."""
    sys.path.insert(0, str(ROOT))
    import torch
    import torch.distributed as dist

    from flygym_amd.sharding import ObsGather, shard_range

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    first, last = shard_range(total_worlds, rank, world_size)
    n_local, nj, n_act = last - first, 66, 42
    g = ObsGather(n_local, nj, n_act, "cpu", total_worlds=total_worlds)
    assert g.world_size == world_size and g.obs_dim == 270

    def fields(tick):          # deterministic per (global world, tick): what a rank's engine would hold after that tick
        w = torch.arange(first, last, dtype=torch.float32)[:, None]
        qpos = w * 1000 + tick * 10 + torch.arange(73, dtype=torch.float32)[None, :] * 0.01
        qvel = -(w * 1000 + tick * 10) + torch.arange(72, dtype=torch.float32)[None, :] * 0.01
        force = w + tick + torch.arange(48, dtype=torch.float32)[None, :] * 0.5
        sens = w * 2 + tick + torch.arange(96, dtype=torch.float32)[None, :] * 0.25
        return qpos, qvel, force, sens

    rows = g.rows()
    seen = []
    live = [t.clone() for t in fields(0)]
    for tick in range(5):
        for dst, src in zip(live, fields(tick)):
            dst.copy_(src)                                   # the "kernel" of this tick writes the fields in place
        k = g.tick(*live)
        assert k == tick
        if tick >= 1:                                        # consume the PREVIOUS tick while this one is in flight
            seen.append(g.wait(tick - 1)[rows].clone())
    seen.append(g.wait()[rows].clone())
    g.drain()
    if rank == 0:
        np.save(tmp, torch.stack(seen).numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("total_worlds", [12, 7])            # equal shards, and shards that differ by one (padding)
def test_two_rank_double_buffered_obs_gather(tmp_path, total_worlds):
    import torch
    import torch.multiprocessing as mp

    out = tmp_path / "gather.npy"
    port = 31500 + (os.getpid() + total_worlds) % 2000
    mp.spawn(_gather_worker, args=(2, port, str(out), total_worlds), nprocs=2, join=True)
    got = np.load(out)
    assert got.shape == (5, total_worlds, 270)
    w = np.arange(total_worlds, dtype=np.float32)[:, None]
    for tick in range(5):
        qpos = w * 1000 + tick * 10 + np.arange(73, dtype=np.float32)[None, :] * np.float32(0.01)
        qvel = -(w * 1000 + tick * 10) + np.arange(72, dtype=np.float32)[None, :] * np.float32(0.01)
        force = w + tick + np.arange(48, dtype=np.float32)[None, :] * np.float32(0.5)
        sens = w * 2 + tick + np.arange(96, dtype=np.float32)[None, :] * np.float32(0.25)
        want = np.concatenate([qpos[:, 7:73], qvel[:, 6:72], force[:, :42], sens], axis=1)
        np.testing.assert_array_equal(got[tick], want.astype(np.float32))


def test_bench_entry_self_spawns_and_shards(monkeypatch):
    """`python bench.py --gpus N` must start N ranks itself: check the launcher command it builds and
    the strong / weak shard arithmetic, without a GPU."""
    import importlib

    sys.path.insert(0, str(ROOT))
    bench = importlib.import_module("bench")
    calls = {}

    def fake_run(cmd, env=None):
        calls["cmd"], calls["env"] = cmd, env
        class R: returncode = 0
        return R()

    monkeypatch.setattr(bench.subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "20", "--warmup", "5"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 0
    cmd = calls["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "8", "--steps", "20", "--warmup", "5"] and cmd[-7].endswith("bench.py")
    assert calls["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    # strong scaling: 4096 worlds over 8 ranks = 512 each, contiguous
    spans = [shard_range(4096, r, 8) for r in range(8)]
    assert spans[0] == (0, 512) and spans[-1] == (3584, 4096)


def test_bench_traffic_and_flop_models():
    """bench.py's derived roofline figures: the HBM traffic of a launch scales from the committed per-world fit
    (profiles/hbm_traffic.json: bytes per launch + bytes per step), the algorithmic FLOP count grows with contacts and
    solver iterations and stays near 0.14 MFLOP per fly-step for the measured walking averages."""
    import importlib
    import json

    sys.path.insert(0, str(ROOT))
    bench = importlib.import_module("bench")
    args = bench.parse_args([])
    rec = json.loads((ROOT / "profiles" / "hbm_traffic.json").read_text())
    t50, issue = bench.traffic_model(4096, 50, args)
    t20, _ = bench.traffic_model(4096, 20, args)
    assert t50 == pytest.approx(rec["traffic_bytes_per_launch"], rel=1e-6)          # the profiled point itself
    assert t50 - t20 == pytest.approx(30 * 4096 * rec["per_world_per_step_bytes"], rel=1e-9)
    # the 168-byte control-table row, plus the state of the chunk hand-offs a longer launch adds: 5.1 KB per world and hand-off
    # (320 granules of 8 bytes, written and read).  Round 5's plans (20 steps in four chunks, 50 in five): one more per 30 steps,
    # 343 B per step; round 6's (20 steps in three chunks): two more per 30 steps, 518 B per step
    assert 160 <= rec["per_world_per_step_bytes"] <= 600
    assert issue["valu_cycles_per_inst"] < 2.0 and issue["valu_insts_per_env_step"] > 5000
    args_terrain = bench.parse_args(["--terrain", "gapped"])
    assert bench.traffic_model(4096, 50, args_terrain) == (None, None)              # only the profiled workload is claimed
    f0 = bench.algorithmic_flops(72, 49, 0.0, 0.0)
    f1 = bench.algorithmic_flops(72, 49, 5.7, 3.4)
    assert f0 < f1 and 1.0e5 < f1 < 1.8e5


def test_shard_plan_fills_gpus_to_residency():
    """Strong scaling of a small batch: a launch below a GPU's residency is pure latency, so `fill` uses
    ceil(total / resident) ranks and leaves the others without worlds; `spread` is shard_range."""
    from flygym_amd.sharding import resident_worlds, shard_plan

    assert resident_worlds(72) == 2048 and resident_worlds(48) == 2048 and resident_worlds(132) == 2048 and resident_worlds(210) == 1280
    assert shard_plan(1024, 8) == [1024] + [0] * 7                       # BASELINE config 5: one GPU holds them all
    assert shard_plan(4096, 8) == [2048, 2048] + [0] * 6
    assert shard_plan(5000, 8) == [1667, 1667, 1666] + [0] * 5
    assert shard_plan(8 * 4096, 8) == [4096] * 8                          # config 4: every GPU is over-subscribed anyway
    assert shard_plan(4096, 8, resident=1792) == [1366, 1365, 1365] + [0] * 5
    assert shard_plan(1024, 8, policy="spread") == [128] * 8
    assert shard_plan(7, 2, policy="spread") == [4, 3]
    for total, ws in [(1, 8), (2047, 3), (2049, 3), (100000, 8)]:
        sizes = shard_plan(total, ws)
        assert sum(sizes) == total and len(sizes) == ws and sizes[0] > 0
        assert all(a >= b for a, b in zip(sizes, sizes[1:]))             # active ranks first
    with pytest.raises(ValueError):
        shard_plan(0, 8)
    with pytest.raises(ValueError):
        shard_plan(10, 2, policy="other")


def _idle_rank_worker(rank, world_size, port, tmp):
    """A `fill` plan over gloo: rank 0 holds all five worlds, rank 1 none — it still joins every all-gather."""
    sys.path.insert(0, str(ROOT))
    import torch
    import torch.distributed as dist

    from flygym_amd.sharding import ObsGather, shard_plan

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    sizes = shard_plan(5, world_size, resident=2048)
    assert sizes == [5, 0]
    n_local = sizes[rank]
    g = ObsGather(n_local, 66, 42, "cpu", total_worlds=5, shard_sizes=sizes)
    assert g.n_max == 5
    w = torch.arange(n_local, dtype=torch.float32)[:, None]
    got = []
    for tick in range(3):
        g.tick(w + tick + torch.zeros((n_local, 73)), -w - tick + torch.zeros((n_local, 72)), w * 2 + torch.zeros((n_local, 48)),
               w * 3 + tick + torch.zeros((n_local, 96)))
        got.append(g.wait()[g.rows()].clone())
    g.drain()
    if rank == 1:                                                # the idle rank sees rank 0's worlds
        np.save(tmp, torch.stack(got).numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_obs_gather_with_an_idle_rank(tmp_path):
    import torch.multiprocessing as mp

    out = tmp_path / "idle.npy"
    mp.spawn(_idle_rank_worker, args=(2, 33500 + os.getpid() % 2000, str(out)), nprocs=2, join=True)
    got = np.load(out)
    assert got.shape == (3, 5, 270)
    w = np.arange(5, dtype=np.float32)
    for tick in range(3):
        np.testing.assert_array_equal(got[tick][:, 0], w + tick)
        np.testing.assert_array_equal(got[tick][:, 66], -w - tick)
        np.testing.assert_array_equal(got[tick][:, 132], w * 2)
        np.testing.assert_array_equal(got[tick][:, 174], w * 3 + tick)
