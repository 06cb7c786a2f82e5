"""World-size-2 GPU test (NCCL = RCCL) of the multi-GPU path over the REAL engine: ShardedBalatroVecEnv around BalatroVecEnv, one
process per GPU, for the bench workload (BASELINE configs[2]) and a configs[3]-style one (card states + jokers + consumables).  Skipped unless two GPUs are visible (the build pool's boxes have one; the driver's 8-GPU node runs it)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _worker(rank, world, port, total, T, kind, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world)
    from balatro_gym_amd.sharded import ShardedBalatroVecEnv
    from balatro_gym_amd.vec_env import RowBuffers
    from tests.helpers import apply_sharded_workload, sharded_workload
    wl = sharded_workload(kind, total)
    env = ShardedBalatroVecEnv(total, wl["seeds"], device=rank, **wl["env_kwargs"])
    apply_sharded_workload(env.local, wl, env.lo, env.hi)
    # the gather WITHOUT a collective, ACROSS devices: every rank's engine writes its current records into the OTHER GPU's buffer through a CUDA IPC
    # mapping (non-temporal stores over xGMI into the peer's HBM, which the peer then reads through its own L2) -- compared byte for byte with the
    # RCCL all_gather of the same rows, for two calls, exactly as tests/test_sharded_one_gpu.py does with two processes on one device
    peer_ok = env.enable_peer_gather()
    rb = RowBuffers(env.hi - env.lo, env.local.device, steps=T)
    env.rollout(T, policy=2, policy_seed=5, obs_buffers=rb)
    rec = env.gather_records(rb.rows[T - 1])
    torch.cuda.synchronize()
    dist.barrier()   # every rank's launch has completed: every buffer is whole
    peer_equal = bool(torch.equal(env.peer_records, rec)) if peer_ok else None
    env.local.observe()
    flat = env.gather_obs()
    st = env.local.stats()
    dist.barrier()
    rb2 = RowBuffers(env.hi - env.lo, env.local.device, steps=7)   # a second, shorter call: the buffers follow the LAST launch of every call
    env.rollout(7, policy=2, policy_seed=5, t0=T, obs_buffers=rb2)
    rec2 = env.gather_records(rb2.rows[6])
    torch.cuda.synchronize()
    dist.barrier()
    peer_equal2 = bool(torch.equal(env.peer_records, rec2)) and not bool(torch.equal(rec, rec2)) if peer_ok else None
    q.put((rank, rec.cpu().numpy(), flat.cpu().numpy(), st, peer_ok, peer_equal, peer_equal2))
    dist.barrier()
    env.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("kind", ["configs2", "configs3"])
def test_two_gpu_sharding_matches_one_gpu(kind):
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    import torch.multiprocessing as mp
    from balatro_gym_amd import BalatroVecEnv
    from balatro_gym_amd.vec_env import RowBuffers
    from tests.helpers import apply_sharded_workload, sharded_workload
    total, world, T = 512, 2, 64
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, T, kind, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    # the same envs on ONE GPU
    wl = sharded_workload(kind, total)
    env = BalatroVecEnv(total, wl["seeds"], device=0, **wl["env_kwargs"])
    apply_sharded_workload(env, wl, 0, total)
    rb = RowBuffers(total, env.device, steps=T)
    env.rollout(T, policy=2, policy_seed=5, obs_buffers=rb)
    want = rb.rows[T - 1].cpu().numpy()
    st1 = env.stats()
    env.close()
    half = total // world
    assert all(r[4] for r in results), "peer-mapped gather buffers could not be set up between the two GPUs (CUDA IPC / peer access)"
    assert all(r[5] and r[6] for r in results), "peer-written records (xGMI stores into the other GPU's buffer) differ from the RCCL all_gather of the same rows"
    for rank, rec, flat, st, *_ in results:
        assert rec.shape == (world, half, 352)
        assert np.array_equal(rec.reshape(total, 352), want), rank   # every rank holds every env's current record
    for k in ("steps", "episodes", "plays", "score_sum"):
        assert sum(r[3][k] for r in results) == st1[k], k
    assert results[0][3]["reward_bits"] ^ results[1][3]["reward_bits"] == st1["reward_bits"]
