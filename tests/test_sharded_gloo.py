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

    def obs_flat_bytes(self, n):
        from balatro_gym_amd.vec_env import obs_flat_bytes
        return obs_flat_bytes(n)

    def close(self):
        pass


class RowsVecEnv:
    """Stand-in for BalatroVecEnv(obs_layout="rows"): `obs_flat` is n records of 384 bytes (here: the GLOBAL env index in every record's
    first two bytes, a running byte pattern behind it), `obs_flat_bytes(n)` = 384 n -- what the sharded gather pads to."""
    STRIDE = 384

    def __init__(self, n, seeds, **kw):
        rows = torch.zeros((n, self.STRIDE), dtype=torch.uint8)
        for i, s in enumerate(seeds):
            g = s - 5000
            rows[i, 0], rows[i, 1] = g & 0xff, g >> 8
            rows[i, 2:352] = torch.arange(350, dtype=torch.int64).add(g).remainder(251).to(torch.uint8)
        self.obs_flat = rows.reshape(-1)

    def obs_flat_bytes(self, n):
        return n * self.STRIDE

    def close(self):
        pass


def _worker_uneven(rank, world, port, total, q, layout):
    """Uneven shards (total % world != 0): every rank must pad to the SAME size in the layout of its local env."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from balatro_gym_amd.sharded import ShardedBalatroVecEnv
    seeds = [5000 + i for i in range(total)]
    env = ShardedBalatroVecEnv(total, seeds, local_env_factory=RowsVecEnv if layout == "rows" else OracleVecEnv)
    gathered = env.gather_obs()
    q.put((rank, env.lo, env.hi, gathered.numpy().copy(), env.local.obs_flat.numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


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


@pytest.mark.parametrize("layout", ["rows", "keys"])
def test_two_rank_uneven_shards_pad_to_one_size(layout):
    """total = 13 over 2 ranks = 7 + 6 envs.  With obs_layout "rows" the flat buffer is 384 bytes per env, which a per-key size formula
    does not describe: rank 0 padded to 7 x 384 and rank 1 to something else, and all_gather_into_tensor ran with mismatched sizes.  Both
    layouts: every rank gathers [world, pad] with pad = the LOCAL layout's size of the biggest shard, own bytes first, zeros behind."""
    total, world = 13, 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_uneven, args=(r, world, port, total, q, layout)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    from balatro_gym_amd.vec_env import obs_flat_bytes
    pad = 7 * 384 if layout == "rows" else obs_flat_bytes(7)
    flats = {rank: local for rank, lo, hi, g, local in results}
    for rank, lo, hi, gathered, local in results:
        assert (hi - lo) == (7 if rank == 0 else 6)
        assert gathered.shape == (world, pad)
        for r in range(world):
            nb = flats[r].size
            assert np.array_equal(gathered[r][:nb], flats[r]) and not gathered[r][nb:].any(), (rank, r)
    if layout == "rows":   # global env indexes arrive in order
        ids = [int(results[0][3][r].reshape(-1, 384)[i, 0]) for r in range(world) for i in range(7 if r == 0 else 6)]
        assert ids == list(range(13))
