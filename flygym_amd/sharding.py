"""Multi-GPU layout: worlds are independent, so they shard contiguously across ranks with no
exchange inside the physics step; the only collective is an all-gather of the observation block
once per control tick (RCCL over xGMI on MI355X, ``backend="nccl"``; ``gloo`` in CPU tests).
The reference is single-GPU (``src/flygym/warp/utils.py:192-202``); this is new functionality."""

from __future__ import annotations

__all__ = ["shard_range", "shard_plan", "resident_worlds", "gather_observations", "ObsGather", "OBS_LAYOUT"]

# joint angles, joint velocities, position-actuator forces, 6 x 16 contact-sensor floats
OBS_LAYOUT = (("joint_angles", 66), ("joint_velocities", 66), ("actuator_forces", 42), ("contact", 96))


def shard_range(total_worlds: int, rank: int, world_size: int) -> tuple[int, int]:
    """Contiguous [first, last) world range of ``rank``; sizes differ by at most one."""
    if not 0 <= rank < world_size:
        raise ValueError("rank out of range")
    base, extra = divmod(total_worlds, world_size)
    first = rank * base + min(rank, extra)
    return first, first + base + (1 if rank < extra else 0)


def resident_worlds(nv: int) -> int:
    """Worlds one MI355X steps at once (one wavefront per world; 256 CUs x the flies a CU holds — register- or
    LDS-limited, ``scripts/kernel_stats.py``): 8 per CU for the leg skeletons (nv <= 72) and ALL_BIOLOGICAL (nv 132; 7 until
    round 3's commit aab7290), 5 for ALL_POSSIBLE (nv 210) and the general-tree kernels' 4 / 3 rounded to the smaller
    figure.  The library asks the occupancy API for the kernel it will launch (``nmf_batch_create``); this table is the
    host-side estimate used before a batch exists."""
    return 2048 if nv <= 132 else 1280 if nv <= 210 else 768


def shard_plan(total_worlds: int, world_size: int, resident: int = 2048, policy: str = "fill") -> list[int]:
    """Worlds per rank for a FIXED total (strong scaling).

    A launch with fewer worlds than a GPU holds at once is pure latency: a step takes as long for 128 worlds as for 2048
    (one wave per world, ~50-85 us per step whatever the count), so spreading a small batch over more GPUs buys little
    and adds the observation exchange.  BASELINE config 5 is the case in point: 1024 flies on 8 GPUs = 128 per GPU ->
    measured 2.08 M env-steps/s per GPU = 16.7 M on eight, where ONE GPU steps all 1024 at 15.1 M (same workload,
    scripts/gpu_config5_sweep.sh): seven more GPUs for +10 %.

    ``policy="fill"`` (default): use ``min(world_size, ceil(total / resident))`` ranks — fill a GPU to its residency
    before taking the next — and split the worlds evenly over those; the remaining ranks get 0 worlds (they stay in the
    collectives and idle).  ``policy="spread"``: every rank gets a share (``shard_range``), whatever its size."""
    if total_worlds <= 0 or world_size <= 0:
        raise ValueError("need positive total_worlds and world_size")
    if policy == "spread":
        return [b - a for a, b in (shard_range(total_worlds, r, world_size) for r in range(world_size))]
    if policy != "fill":
        raise ValueError("policy must be 'fill' or 'spread'")
    n_active = min(world_size, max(1, -(-total_worlds // max(1, resident))))
    sizes = [b - a for a, b in (shard_range(total_worlds, r, n_active) for r in range(n_active))]
    return sizes + [0] * (world_size - n_active)


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


class ObsGather:
    """The one exchange of the multi-GPU path: per control tick every rank packs its observation block
    ``[n_local, 2 nj + n_act + 96]`` (joint angles, joint velocities, position-actuator forces, contact sensors) and
    all-gathers it to every rank.  Double-buffered and asynchronous: the gather of tick ``k`` runs on the communication
    stream (RCCL over xGMI; ``gloo`` on CPU) while the stepping kernel of tick ``k + 1`` already runs; buffer pair
    ``k & 1`` is reused at tick ``k + 2`` only after its gather has completed.

    Ranks may own different numbers of worlds (``shard_range`` sizes differ by at most one): blocks are padded to
    ``n_max`` rows and :meth:`rows` maps a gathered buffer back to the global world order.
    """

    def __init__(self, n_local: int, nj: int, n_act: int, device, *, total_worlds: int | None = None, group=None,
                 shard_sizes: list[int] | None = None, packer=None):
        """``shard_sizes``: worlds per rank when they are not ``shard_range``'s (``shard_plan``: some ranks may hold none).
        ``packer``: ``HIPSimulation.pack_observations`` — packs the block in one launch instead of four tensor copies."""
        self._packer = packer
        import torch
        import torch.distributed as dist

        self._torch, self._dist, self._group = torch, dist, group
        self.distributed = dist.is_available() and dist.is_initialized()
        self.world_size = dist.get_world_size(group) if self.distributed else 1
        self.rank = dist.get_rank(group) if self.distributed else 0
        self.n_local, self.nj, self.n_act = int(n_local), int(nj), int(n_act)
        self.obs_dim = 2 * self.nj + self.n_act + 96
        self.total_worlds = int(total_worlds) if total_worlds is not None else self.n_local * self.world_size
        if shard_sizes is not None:
            if len(shard_sizes) != self.world_size or sum(shard_sizes) != self.total_worlds or shard_sizes[self.rank] != self.n_local:
                raise ValueError("shard_sizes must list every rank's worlds and add up to total_worlds")
            self.shard_sizes = [int(x) for x in shard_sizes]
        else:
            self.shard_sizes = [b - a for a, b in (shard_range(self.total_worlds, r, self.world_size) for r in range(self.world_size))]
        self.n_max = max(self.shard_sizes)
        if self.n_local > self.n_max:
            raise ValueError("n_local exceeds the padded shard size")
        self.local = [torch.zeros((self.n_max, self.obs_dim), dtype=torch.float32, device=device) for _ in range(2)]
        self.full = [torch.zeros((self.world_size * self.n_max, self.obs_dim), dtype=torch.float32, device=device)
                     for _ in range(2)]
        self.pending = [None, None]
        self.ticks = 0

    def pack(self, out, qpos, qvel, actuator_force, sensordata):
        """Columns in OBS_LAYOUT order from the engine's raw fields (views; device-side copies only)."""
        nj, na, n = self.nj, self.n_act, self.n_local
        if self._packer is not None and n > 0:
            self._packer(out, na)
            return out
        out[:n, 0:nj] = qpos[:, 7:7 + nj]
        out[:n, nj:2 * nj] = qvel[:, 6:6 + nj]
        out[:n, 2 * nj:2 * nj + na] = actuator_force[:, :na]
        out[:n, 2 * nj + na:] = sensordata
        return out

    def tick(self, qpos, qvel, actuator_force, sensordata):
        """Pack this tick's block and start its gather; returns the tick index (buffer pair = index & 1)."""
        k = self.ticks
        slot = k & 1
        if self.pending[slot] is not None:
            self.pending[slot].wait()            # the gather that used this buffer pair two ticks ago
            self.pending[slot] = None
        ol = self.pack(self.local[slot], qpos, qvel, actuator_force, sensordata)
        if self.distributed:
            self.pending[slot] = self._dist.all_gather_into_tensor(self.full[slot], ol, group=self._group, async_op=True)
        else:
            self.full[slot].copy_(ol)
        self.ticks = k + 1
        return k

    def wait(self, k: int | None = None):
        """Block the current stream on the gather of tick ``k`` (default: the latest) and return its buffer
        ``[world_size * n_max, obs_dim]``."""
        k = self.ticks - 1 if k is None else k
        if k < 0 or k < self.ticks - 2:
            raise ValueError("only the two most recent ticks are still buffered")
        slot = k & 1
        if self.pending[slot] is not None:
            self.pending[slot].wait()
            self.pending[slot] = None
        return self.full[slot]

    def drain(self):
        for slot in (0, 1):
            if self.pending[slot] is not None:
                self.pending[slot].wait()
                self.pending[slot] = None

    def rows(self):
        """Index tensor selecting, from a gathered buffer, the rows of the real worlds in global world order."""
        idx = []
        for r in range(self.world_size):
            idx.extend(range(r * self.n_max, r * self.n_max + self.shard_sizes[r]))
        return self._torch.as_tensor(idx, dtype=self._torch.long, device=self.full[0].device)
