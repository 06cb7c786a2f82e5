"""Sharding independent envs over the GPUs of a node: one process per GPU (torch.distributed, backend "nccl" = RCCL).

Envs never interact, so the data path has NO collective: rank r owns the contiguous env range shard_range(total, W, r)
and steps it locally.  The only exchange is the OPTIONAL gather of the observation tensor for a learner that wants all
shards (`gather_obs`): one all_gather of the flat observation buffer (all 31 keys live in one allocation, so one
collective moves every key).  On MI355X the 8 GPUs are fully connected by xGMI, so RCCL can move each rank's shard on
its own link; at 65 536 envs x 330 B the whole observation is 21.6 MB (2.7 MB per GPU).
"""
from __future__ import annotations

from typing import Callable, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def shard_range(total: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous split; the first (total % world) ranks own one extra env."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class ShardedBalatroVecEnv:
    """total_envs games split over the process group; every rank drives its own shard on its own GPU."""

    def __init__(self, total_envs: int, seeds: Sequence[int], *, rank: Optional[int] = None, world: Optional[int] = None,
                 local_env_factory: Optional[Callable] = None, group=None, **env_kwargs):
        self.group = group
        self.rank = dist.get_rank(group) if rank is None else rank
        self.world = dist.get_world_size(group) if world is None else world
        self.total_envs = int(total_envs)
        if len(seeds) != total_envs:
            raise ValueError("seeds must list one master seed per GLOBAL env")
        self.lo, self.hi = shard_range(self.total_envs, self.world, self.rank)
        if local_env_factory is None:
            from .vec_env import BalatroVecEnv
            local_env_factory = BalatroVecEnv
        self.local = local_env_factory(self.hi - self.lo, list(seeds[self.lo:self.hi]), **env_kwargs)
        self._gathered = None

    @property
    def env_index0(self) -> int:
        return self.lo

    def reset(self, **kw):
        return self.local.reset(**kw)

    def step(self, actions_local: torch.Tensor):
        return self.local.step(actions_local)

    def rollout(self, steps: int, **kw):
        kw.setdefault("env_index0", self.lo)
        return self.local.rollout(steps, **kw)

    def gather_obs(self) -> torch.Tensor:
        """all_gather of the flat observation buffer -> uint8 [world, shard_bytes] (smaller shards are zero padded)."""
        flat = self.local.obs_flat
        nbytes = torch.tensor([flat.numel()], dtype=torch.int64, device=flat.device)
        if self.total_envs % self.world:
            dist.all_reduce(nbytes, op=dist.ReduceOp.MAX, group=self.group)
        n = int(nbytes.item())
        if flat.numel() != n:
            buf = torch.zeros(n, dtype=torch.uint8, device=flat.device)
            buf[:flat.numel()] = flat
            flat = buf
        if self._gathered is None or self._gathered.numel() != self.world * n:
            self._gathered = torch.empty(self.world * n, dtype=torch.uint8, device=flat.device)
        dist.all_gather_into_tensor(self._gathered, flat.contiguous(), group=self.group)
        return self._gathered.view(self.world, n)

    def close(self):
        self.local.close()
