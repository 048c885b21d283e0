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
