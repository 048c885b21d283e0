"""Multi-GPU layout: worlds are independent, so they shard contiguously across ranks with no
exchange inside the physics step; the only collective is an all-gather of the observation block
once per control tick (RCCL over xGMI on MI355X, ``backend="nccl"``; ``gloo`` in CPU tests).
The reference is single-GPU (``src/flygym/warp/utils.py:192-202``); this is new functionality."""

from __future__ import annotations

__all__ = ["shard_range", "gather_observations", "OBS_LAYOUT"]

# joint angles, joint velocities, position-actuator forces, 6 x 16 contact-sensor floats
OBS_LAYOUT = (("joint_angles", 66), ("joint_velocities", 66), ("actuator_forces", 42), ("contact", 96))


def shard_range(total_worlds: int, rank: int, world_size: int) -> tuple[int, int]:
    """Contiguous [first, last) world range of ``rank``; sizes differ by at most one."""
    if not 0 <= rank < world_size:
        raise ValueError("rank out of range")
    base, extra = divmod(total_worlds, world_size)
    first = rank * base + min(rank, extra)
    return first, first + base + (1 if rank < extra else 0)


def gather_observations(obs_local, out=None):
    """All-gather ``(n_local, obs_dim)`` blocks into ``(world_size * n_local, obs_dim)`` on every
    rank (equal ``n_local`` on all ranks).  No-op copy when not running distributed."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return obs_local if out is None else out.copy_(obs_local)
    ws = dist.get_world_size()
    if out is None:
        out = torch.empty((ws * obs_local.shape[0],) + tuple(obs_local.shape[1:]), dtype=obs_local.dtype,
                          device=obs_local.device)
    dist.all_gather_into_tensor(out, obs_local.contiguous())
    return out
