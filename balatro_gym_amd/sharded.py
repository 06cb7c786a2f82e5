"""Sharding independent envs over the GPUs of a node: one process per GPU (torch.distributed, backend "nccl" = RCCL).

Envs never interact, so the data path has NO collective: rank r owns the contiguous env range shard_range(total, W, r)
and steps it locally.  The one exchange of the design is the gather of the CURRENT observation of every env, for whoever
needs all shards in one place (a central evaluator, a logger, a learner that does not train shard-locally):
  * `gather_obs()`      after reset() / step(): one all_gather of the flat per-key observation buffer (all 31 keys live in
                        one allocation, so one collective moves every key);
  * `gather_records(r)` after a fused rollout with packed records: one all_gather of a [n, 352] record row (normally the
                        last row of the launch, `rows[T - 1]`) -- what bench.py times at N > 1.
The [T, N] record history of a fused rollout is NOT gathered: at 65 536 envs per GPU it is produced at ~1.8 TB/s per GPU,
more than the ~1.07 TB/s of xGMI a GPU has (7 links x 153 GB/s); learners consume their own shard (data-parallel PPO
all-reduces gradients, not observations).  On MI355X the 8 GPUs are fully connected by xGMI, so RCCL moves each rank's shard
on its own link: 65 536 x 352 B = 23 MB per GPU and launch, ~0.15 ms per link, beside a ~4 ms launch.
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


def all_gather_bytes(out: torch.Tensor, inp: torch.Tensor, group=None) -> None:
    """all_gather_into_tensor(out, inp).  RCCL ("nccl") moves device buffers directly over xGMI.  The "gloo" backend -- CPU tests, and
    the two-ranks-on-one-GPU test of the real engine -- gets device tensors staged through host memory."""
    if inp.is_cuda and dist.get_backend(group) == "gloo":
        h_in = inp.cpu()
        h_out = torch.empty(out.numel(), dtype=out.dtype)
        dist.all_gather_into_tensor(h_out, h_in.view(-1), group=group)
        out.view(-1).copy_(h_out)
        return
    dist.all_gather_into_tensor(out, inp, group=group)


def setup_peer_gather(env, rank: int, world: int, group=None) -> Optional[torch.Tensor]:
    """The gather WITHOUT a collective: every rank allocates uint8 [world, n, 352], hands its CUDA IPC handle to the others (all_gather_object: a few
    hundred bytes, once), opens theirs, and tells its engine (BalatroVecEnv.set_gather_peers): from then on the last launch of every packed-record rollout
    writes every env's current record into slot [rank] of ALL buffers from its copy-out -- peer-mapped stores over xGMI while the launch runs.  After
    the rollout and a barrier between the ranks, the returned tensor holds the current record of every env of the job.  Returns None (on every rank)
    when the mapping is not available on some rank: the caller then uses the RCCL all_gather (`gather_records`)."""
    from torch.multiprocessing.reductions import reduce_tensor
    n = env.num_envs
    ok, mine, peers, err = True, None, [None] * world, ""
    try:
        mine = torch.zeros((world, n, 352), dtype=torch.uint8, device=env.device)
        fn, args = reduce_tensor(mine)
    except Exception as ex:  # noqa: BLE001
        ok, fn, args, err = False, None, None, repr(ex)
    handles = [None] * world
    dist.all_gather_object(handles, (ok, fn, args, n), group=group)
    if ok and all(h[0] and h[3] == n for h in handles):
        try:
            for r in range(world):
                peers[r] = mine if r == rank else handles[r][1](*handles[r][2])
                if tuple(peers[r].shape) != (world, n, 352):
                    raise RuntimeError("peer gather buffer has another shape")
        except Exception as ex:  # noqa: BLE001
            ok, err = False, repr(ex)
    else:
        ok = False
    if ok:
        try:   # (the library checks that this device may write every buffer -- peer access to the device it lives on -- and refuses otherwise)
            env.set_gather_peers(peers, rank)
        except Exception as ex:  # noqa: BLE001
            ok, err = False, repr(ex)
    flags = [None] * world
    dist.all_gather_object(flags, (ok, err), group=group)
    if not all(f[0] for f in flags):   # one rank could not: NO rank writes peers (a rank that kept writing would write into buffers nobody reads)
        try:
            env.set_gather_peers([], rank)
        except Exception:  # noqa: BLE001
            pass
        # every rank comes through here together (the flags were all-gathered): no engine writes peers any more once all have passed the barrier --
        # only then may `mine` (which the others hold IPC mappings of) and the mappings of theirs go out of scope
        dist.barrier(group=group)
        return None
    return mine


def teardown_peer_gather(env, rank: int, group=None) -> None:
    """Switch the peer-written gather off on EVERY rank (collective): the engines stop writing, a barrier, and only then does a rank let go of the
    buffer its peers have mapped (`BalatroVecEnv.set_gather_peers([])` drops the handle's references; the caller drops the tensor
    `setup_peer_gather` returned AFTER this call)."""
    keep = getattr(env, "_gather_keep", None)   # our buffer and the peers' mappings: alive until every rank has stopped writing
    env.set_gather_peers([], rank)
    torch.cuda.synchronize(env.device)
    dist.barrier(group=group)
    del keep


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
        self._pad_bytes = None
        self.peer_records = None

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
        # the padded shard size is a function of the split alone (the first total % world ranks own one env more, and every env has the
        # same number of observation bytes): computed on the host, once -- a MAX all-reduce + `.item()` here was a host sync per call
        if self._pad_bytes is None:
            biggest = max(shard_range(self.total_envs, self.world, r)[1] - shard_range(self.total_envs, self.world, r)[0] for r in range(self.world))
            # the LOCAL env says how many bytes `biggest` envs take in ITS layout (per-key arrays, or 384-byte records with obs_layout="rows"):
            # every rank then pads to the same size whatever the layout -- a per-key formula here gave uneven "rows" shards different sizes
            flat_bytes = getattr(self.local, "obs_flat_bytes", None)
            if flat_bytes is None:
                raise TypeError("the local env must expose obs_flat_bytes(n): bytes of the flat observation buffer of n envs in its layout")
            self._pad_bytes = int(flat_bytes(biggest))
            if flat.numel() > self._pad_bytes:
                raise ValueError(f"local observation buffer ({flat.numel()} bytes) exceeds obs_flat_bytes({biggest}) = {self._pad_bytes}")
        n = self._pad_bytes
        if flat.numel() != n:
            buf = torch.zeros(n, dtype=torch.uint8, device=flat.device)
            buf[:flat.numel()] = flat
            flat = buf
        if self._gathered is None or self._gathered.numel() != self.world * n:
            self._gathered = torch.empty(self.world * n, dtype=torch.uint8, device=flat.device)
        all_gather_bytes(self._gathered, flat.contiguous(), group=self.group)
        return self._gathered.view(self.world, n)

    def enable_peer_gather(self) -> bool:
        """Switch the record gather to peer-mapped writes from the engine (setup_peer_gather).  True: after every packed-record `rollout` and a
        barrier between the ranks, `peer_records` holds [world, n, 352]; False: not available here, `gather_records` (RCCL) is the way."""
        if self.total_envs % self.world:
            return False
        self.peer_records = setup_peer_gather(self.local, self.rank, self.world, self.group)
        return self.peer_records is not None

    def gather_records(self, rows: torch.Tensor) -> torch.Tensor:
        """all_gather of one packed-record row of this shard (uint8 [n, 352], e.g. `RowBuffers.rows[T - 1]`) ->
        uint8 [world, n, 352].  Shards must have equal sizes (total_envs divisible by the world size)."""
        if self.total_envs % self.world:
            raise ValueError("gather_records needs equal shards")
        rows = rows.contiguous()
        out = torch.empty((self.world,) + tuple(rows.shape), dtype=rows.dtype, device=rows.device)
        all_gather_bytes(out.view(-1), rows.view(-1), group=self.group)
        return out

    def disable_peer_gather(self) -> None:
        """Back to `gather_records` (collective: every rank calls it)."""
        if self.peer_records is not None:
            teardown_peer_gather(self.local, self.rank, self.group)
            self.peer_records = None

    def close(self):
        self.disable_peer_gather()
        self.local.close()
