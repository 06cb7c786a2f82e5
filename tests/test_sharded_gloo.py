"""World-size-2 CPU test (gloo) of the multi-GPU path: contiguous env sharding, per-rank stepping with the right GLOBAL
env indexes, and the observation all_gather.  The per-rank engine is the CPU oracle here (tests may use it); on the GPU
box the same ShardedBalatroVecEnv wraps BalatroVecEnv."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OracleVecEnv:
    """Minimal stand-in with BalatroVecEnv's surface (obs_flat / step / rollout) on CPU tensors."""

    def __init__(self, n, seeds, **kw):
        from balatro_gym_amd.vec_env import ObsBuffers
        from oracle import pyoracle as po
        self.po = po
        self.envs = [po.OracleEnv(s) for s in seeds]
        self._obs = ObsBuffers(n, torch.device("cpu"))
        self.obs_flat = self._obs.flat
        self._write()

    def _write(self):
        for i, e in enumerate(self.envs):
            o = e.obs()
            for k, t in self._obs.tensors.items():
                t[i] = torch.as_tensor(np.asarray(o[k]))

    def rollout(self, steps, policy=2, policy_seed=0, env_index0=0, t0=0, **kw):
        total = 0.0
        for t in range(steps):
            for i, e in enumerate(self.envs):
                _, r, term, _, _ = e.step(e.policy_action(policy, policy_seed, env_index0 + i, t0 + t))
                total += r
                if term:
                    e.reset()
        self._write()
        return total

    def close(self):
        pass


def _worker(rank, world, port, total, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from balatro_gym_amd.sharded import ShardedBalatroVecEnv, shard_range
    seeds = [5000 + i for i in range(total)]
    env = ShardedBalatroVecEnv(total, seeds, local_env_factory=OracleVecEnv)
    assert (env.lo, env.hi) == shard_range(total, world, rank)
    env.rollout(40, policy=2, policy_seed=3)
    gathered = env.gather_obs()
    # the packed-record gather (bench.py at N > 1): one [n, 352] row per rank, here the rank's env indexes as a marker
    rows = torch.zeros((env.hi - env.lo, 352), dtype=torch.uint8)
    rows[:, 0] = torch.arange(env.lo, env.hi, dtype=torch.uint8)
    rec = env.gather_records(rows)
    assert tuple(rec.shape) == (world, env.hi - env.lo, 352)
    for r in range(world):
        lo_r, hi_r = shard_range(total, world, r)
        assert rec[r, :, 0].tolist() == list(range(lo_r, hi_r)) and not rec[r, :, 1:].any()
    q.put((rank, gathered.numpy().copy(), env.local.obs_flat.numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_matches_single_process():
    total, world = 12, 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single-process reference over the same global env indexes
    sys.path.insert(0, ROOT)
    ref = OracleVecEnv(total, [5000 + i for i in range(total)])
    ref.rollout(40, policy=2, policy_seed=3, env_index0=0)
    from balatro_gym_amd import _native as nat
    from balatro_gym_amd.vec_env import ObsBuffers
    half = total // world
    for rank, gathered, local_flat in results:
        assert gathered.shape[0] == world
        assert np.array_equal(gathered[rank], local_flat)
        for r in range(world):
            shard = ObsBuffers(half, torch.device("cpu"))
            shard.flat.copy_(torch.from_numpy(gathered[r][:shard.flat.numel()]))
            for k in nat.OBS_KEYS:
                want = ref._obs.tensors[k][r * half:(r + 1) * half]
                assert torch.equal(shard.tensors[k], want), (rank, r, k)
