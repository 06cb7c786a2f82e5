"""The N > 1 path over the REAL engine on a ONE-GPU box: two processes, each a BalatroVecEnv shard on device 0 under
ShardedBalatroVecEnv, torch.distributed backend "gloo" (RCCL refuses two ranks on one device; the gather is staged through host
memory, sharded.all_gather_bytes).  Everything else -- shard ranges, env_index0, the record / observation gathers, the statistics --
is the code the 8-GPU job runs.  Also bench.py's own N > 1 branch (fixed-count warm-up, gather inside the timed region, samples)
with two ranks on one GPU, and its refusal to start more ranks than there are GPUs."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _worker(rank, world, port, total, T, kind, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from balatro_gym_amd.sharded import ShardedBalatroVecEnv
    from balatro_gym_amd.vec_env import RowBuffers
    from tests.helpers import apply_sharded_workload, sharded_workload
    wl = sharded_workload(kind, total)
    env = ShardedBalatroVecEnv(total, wl["seeds"], device=0, **wl["env_kwargs"])
    apply_sharded_workload(env.local, wl, env.lo, env.hi)
    # the gather WITHOUT a collective: every rank's engine writes its current records into every rank's buffer (CUDA IPC handles work between two
    # processes on one device as they do between two GPUs); compared below, byte for byte, with the all_gather of the same rows
    peer_ok = env.enable_peer_gather()
    rb = RowBuffers(env.hi - env.lo, env.local.device, steps=T)
    env.rollout(T, policy=2, policy_seed=5, obs_buffers=rb)
    rec = env.gather_records(rb.rows[T - 1])
    torch.cuda.synchronize()
    dist.barrier()   # every rank's launch has completed: every buffer is whole
    if peer_ok:
        assert torch.equal(env.peer_records, rec), "peer-written records differ from the all_gather of the same rows"
    env.local.observe()
    flat = env.gather_obs()
    st = env.local.stats()
    q.put((rank, rec.cpu().numpy(), flat.cpu().numpy(), st, peer_ok))
    dist.barrier()
    # a second, shorter call: the buffers follow the LAST launch of every call
    rb2 = RowBuffers(env.hi - env.lo, env.local.device, steps=7)
    env.rollout(7, policy=2, policy_seed=5, t0=T, obs_buffers=rb2)
    rec2 = env.gather_records(rb2.rows[6])
    torch.cuda.synchronize()
    dist.barrier()
    if peer_ok:
        assert torch.equal(env.peer_records, rec2)
        assert not torch.equal(rec, rec2)
    dist.barrier()
    env.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("kind", ["configs2", "configs3"])
def test_two_ranks_one_gpu_match_one_process(kind):
    import torch.multiprocessing as mp
    from balatro_gym_amd import BalatroVecEnv
    from balatro_gym_amd.vec_env import RowBuffers
    from tests.helpers import apply_sharded_workload, sharded_workload
    total, world, T = 1024, 2, 96
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, T, kind, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=900) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    wl = sharded_workload(kind, total)
    env = BalatroVecEnv(total, wl["seeds"], device=0, **wl["env_kwargs"])
    apply_sharded_workload(env, wl, 0, total)
    rb = RowBuffers(total, env.device, steps=T)
    env.rollout(T, policy=2, policy_seed=5, obs_buffers=rb)
    want = rb.rows[T - 1].cpu().numpy()
    env.observe()
    want_flat = env.obs_flat.cpu().numpy()
    st1 = env.stats()
    env.check()
    env.close()
    half = total // world
    assert all(r[4] for r in results), "peer-mapped gather buffers could not be set up on this box (CUDA IPC between two processes on one device)"
    for rank, rec, flat, st, _ in results:
        assert rec.shape == (world, half, 352)
        assert np.array_equal(rec.reshape(total, 352), want), rank   # every rank holds every env's current record
        assert flat.shape[0] == world
    if kind == "configs3":
        from balatro_gym_amd import _native as nat
        o = nat.ROW_OFFSETS["consumable_count"]
        assert st1["episodes"] > 0 and want[:, o].max() > 0   # (consumables are held: the workload is the one meant)
    # the gathered per-key buffers hold each shard's keys back to back: compare shard 0's keys with the first half of every key of the
    # one-process run through the vec env's own layout
    for k in ("steps", "episodes", "plays", "score_sum"):
        assert sum(r[3][k] for r in results) == st1[k], k
    assert results[0][3]["reward_bits"] ^ results[1][3]["reward_bits"] == st1["reward_bits"]
    assert want_flat.size == results[0][2].shape[1] * world


def _bench(args, timeout=900):
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=timeout,
                          env={k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")})


def test_bench_two_ranks_one_gpu():
    """bench.py --gpus 2 started WITHOUT a launcher: it spawns the ranks itself; both ranks run the fixed-count warm-up, the gather
    of the current record inside the timed region, the repeated samples and the sustained window, and rank 0 prints the line."""
    out = _bench(["--gpus", "2", "--share-gpu", "--dist-backend", "gloo", "--envs-per-gpu", "2048", "--steps", "20", "--warmup", "5",
                  "--samples", "3", "--internal-warmup-launches", "5", "--no-cpu-baseline"])
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    j = json.loads(line)
    assert j["n_gpus"] == 2 and j["config"]["total_envs"] == 4096 and j["steps"] == 20
    assert j["gather"]["in_timed_region"] and j["gather"]["bytes_per_gpu_per_launch"] == 2048 * 352
    # the peer-written buffers were compared with an all_gather of the same rows, byte for byte, on both ranks, before anything was timed
    assert j["gather"]["method"] == "peer" and j["gather"]["verified"] is True, j["gather"]
    assert j["samples"]["n"] == 3 and j["sustained"]["regions"] >= 2
    assert j["value"] > 0


def test_bench_refuses_more_ranks_than_gpus():
    import torch
    n = torch.cuda.device_count() + 1
    out = _bench(["--gpus", str(n), "--steps", "20", "--warmup", "5"], timeout=300)
    assert out.returncode == 2
    assert "GPU(s) are visible" in out.stderr and "Traceback" not in out.stderr
