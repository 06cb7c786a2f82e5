#!/usr/bin/env python3
"""bench.py -- env-steps/sec of the MI355X Balatro step path (BASELINE.json metric), one JSON line on rank 0.

Contract: `python bench.py --gpus N --steps K --warmup W`; for N > 1 the driver launches one rank per GPU with
torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the env).  A "step" is ONE lockstep env step of
the whole per-GPU batch (65 536 envs) through the fused rollout path: counter-hash random policy computed on device,
SAME_STEP auto-reset, the observation of EVERY step written to HBM.  K steps are timed between barrier +
torch.cuda.synchronize() on both sides; value = (envs on all ranks x K) / max-over-ranks time.

Before the W warm-up steps the bench runs an UNTIMED internal warm-up of full-depth launches (~0.5 s: clocks, TLBs, the
look-ahead pipeline in steady state), so a short driver run (`--steps 20 --warmup 5`) measures the same machine state as a
long one.

Workload = BASELINE.json configs[2] (the config the metric is quoted on, fits one GPU): 65 536 envs per GPU, 5 random
jokers per env out of the 51 that complete_joker_effects implements (scorer-level joker chain live), Antes 1-4 cap,
policy: blind 45/46/47 by env index, shop -> 31, otherwise uniform over valid actions.  Envs are independent, so
multi-GPU is pure sharding with no data-path collective ("weak" scaling: 65 536 envs per GPU); the one exchange of the
design -- every rank sees the CURRENT observation record of every env ([N, 352] bytes per GPU, once per launch) -- is inside the
timed region when N > 1 (`gather` in the JSON line): written by the engine's copy-out straight into every rank's buffer (peer-mapped
stores over xGMI, `--gather peer`, the default), or an RCCL all_gather on a side stream (`--gather rccl`, and the fallback).

`value` is the contract's K-step region; `roofline.frac` its algorithmic bytes over its WALL time (the kernel's own HIP-event figure beside it as
`kernel_frac`).  `samples` repeats that same region (default 30 times, each its own launch sequence between two synchronisations of the launch
stream): median / p10 / p90; `sustained` times enough back-to-back repeats of the K-step region WITHOUT a synchronisation in between to span at least
one period of the RNG look-ahead refill.  The rings ask for a refill every 18th launch at 20 steps; beside short launches it is issued IN PIECES (its
scan beside the launch that asked, then one dense kernel over a part of a work list beside each of the next ~17 launches: bg_lib.hip, bg_refill_pieces),
so the single 20-step region of the contract carries its piece like every other launch, `sustained` carries a whole period, and `samples` shows the
spread between the launches beside a cheap piece (shop-stream seeding: ALU) and a dear one (deck shuffles: every lane in its own 2.5 KB MT state).

`python bench.py --gpus N` without torchrun (WORLD_SIZE unset) starts N child ranks itself -- before this process touches a GPU -- and
exits non-zero with a message if fewer than N devices are visible.

Beside the headline the line carries `step_path`: the same workload driven through bg_step (one call per step, actions
from a device tensor: the surface RL code calls, balatro_env_2.py:616-637) and bg_step_many (K steps per call), timed on a
short sample outside the headline's timed region.
"""
from __future__ import annotations

import argparse
import ctypes as C
import glob
import json
import os
import random
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

ENVS_PER_GPU = 65536
OBS_BYTES, IO_BYTES, STATE_BYTES = 330, 10, 192  # A_step(T) = 340 + 384 / T bytes per env-step
# (the packed-record layout writes 352 bytes per env-step, 12 of them padding / the action and terminated flag; the
#  roofline keeps SURVEY's 340 algorithmic bytes)
OBS_ROW_WRITE_BYTES = OBS_BYTES + IO_BYTES        # algorithmic bytes WRITTEN per env-step (the state adds 192 / T)
HBM_PEAK_GBPS = 8000.0                           # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec peak)

IMPLEMENTED = [1, 136, 27, 38, 61, 16, 34, 108, 23, 22, 53, 97, 50, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15,
               131, 132, 133, 134, 135, 48, 128, 122, 72, 140, 31, 39, 40, 41, 101, 124, 26, 33, 104, 147, 118, 119,
               116, 117]
POLICY_CYCLE3 = 2
POLICY_SEED = 20251001
MAX_ANTE = 4


def jokers_for(global_env: int):
    return random.Random(global_env).sample(IMPLEMENTED, 5)


def cpu_baseline(n_envs: int, steps: int, threads: int):
    """The CPU oracle ("port": plain-C restatement of the reference path) on the host cores, same workload/policy."""
    from oracle import pyoracle as po
    L = po.lib()
    envs = []
    for i in range(n_envs):
        e = po.OracleEnv(1000 + i, scorer_jokers=True, max_ante=MAX_ANTE)
        e.set_template_jokers(jokers_for(i))
        envs.append(e)
    per = (n_envs + threads - 1) // threads
    results = [0] * threads

    def work(k):
        lo, hi = k * per, min(n_envs, (k + 1) * per)
        if lo >= hi:
            return
        arr = (C.c_void_p * (hi - lo))(*[envs[i].handle for i in range(lo, hi)])
        rs, ss, ep = C.c_double(), C.c_int64(), C.c_int64()
        results[k] = L.bo_rollout(arr, hi - lo, lo, steps, POLICY_CYCLE3, POLICY_SEED, 0, C.byref(rs), C.byref(ss), C.byref(ep))

    t0 = time.perf_counter()
    ths = [threading.Thread(target=work, args=(k,)) for k in range(threads)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    dt = time.perf_counter() - t0
    return sum(results) / dt, dt


def measured_copy_gbps(dev):
    """What a plain 16-byte-per-lane streaming copy sustains on this GPU (bg_bench_copy: read + written bytes per second), and what a
    plain 16-byte-per-lane FILL sustains (bg_bench_fill: written bytes per second) -- the step engine's traffic is ~80 % stores."""
    import torch
    from balatro_gym_amd import _native as nat
    L = nat.load()
    nbytes = 1 << 30
    src = torch.empty(nbytes, dtype=torch.uint8, device=dev).random_(0, 256)
    dst = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    g = C.c_double()
    with torch.cuda.device(dev):
        rc = L.bg_bench_copy(C.c_void_p(src.data_ptr()), C.c_void_p(dst.data_ptr()), C.c_uint64(nbytes), 20, C.byref(g),
                             C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
    if rc != 0:
        raise RuntimeError(f"bg_bench_copy failed: {L.bg_last_error(None).decode()}")
    gw = C.c_double()
    with torch.cuda.device(dev):
        rc = L.bg_bench_fill(C.c_void_p(dst.data_ptr()), C.c_uint64(nbytes), 20, C.byref(gw),
                             C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
    if rc != 0:
        raise RuntimeError(f"bg_bench_fill failed: {L.bg_last_error(None).decode()}")
    del src, dst
    return float(g.value), float(gw.value)


HBM_PEAK_GUIDE_GBPS = 6290.0                     # MI355X_MICROARCH.md: 6.29 TB/s measured for a float4 copy


def matching_traffic(kernel: str, n: int, fused: float):
    """HBM bytes per launch of the dominant kernel from the PMC counters -- ONLY if a committed measurement of THIS launch
    shape on THIS device code exists (profiles/*_hbm_traffic.json written by tools/hbm_traffic.py from separate rocprofv3
    --pmc FETCH_SIZE / WRITE_SIZE passes of this command, stamped with the library's build signature, bg_build_signature(): sha256 of its sources, flags and hipcc version);
    otherwise null.  Never a value scaled from another shape or taken on another build."""
    from balatro_gym_amd import _native as nat
    sig = nat.device_code_signature()
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_hbm_traffic.json"))):
        try:
            with open(path) as f:
                j = json.load(f)
            if j.get("kernel") == kernel and j.get("device_code_sha") == sig and int(j.get("envs", 0)) == n and int(round(j.get("fused_steps_per_launch", 0))) == int(round(fused)):
                best = (float(j["hbm_bytes_per_launch"]), os.path.basename(path))
        except Exception:
            continue
    return best


def spawn_ranks(n_gpus: int, need_devices: int) -> int:
    """`--gpus N` without a launcher: start N ranks of this script as child processes.  Nothing in THIS process has touched a GPU
    (torch.cuda.device_count() does not initialise HIP on this image), and it never re-executes itself."""
    import socket
    import subprocess
    import torch
    ndev = torch.cuda.device_count()
    if ndev < need_devices:
        print(f"bench.py: --gpus {n_gpus} asked for but only {ndev} GPU(s) are visible; run on a box with {n_gpus} GPUs "
              f"(or under torch.distributed.run with one rank per GPU)", file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n_gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    for pr in procs:
        rc = max(rc, abs(pr.wait()))
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=7440)   # 20 launches of 372 fused steps
    ap.add_argument("--warmup", type=int, default=744)
    ap.add_argument("--envs-per-gpu", type=int, default=ENVS_PER_GPU)
    ap.add_argument("--chunk", type=int, default=0, help="fused steps per rollout call (0 = as many as the rings allow)")
    ap.add_argument("--keep-obs", type=int, default=1, help="write every step's observation to a [chunk, N] buffer")
    ap.add_argument("--obs-layout", choices=["rows", "keys"], default="rows",
                    help="rows: one packed 352-byte record per (step, env) (bg_rollout_rows); keys: one array per key")
    ap.add_argument("--internal-warmup-launches", type=int, default=128,
                    help="untimed full-depth launches before --warmup: a FIXED count (~0.5 s), identical on every rank -- a time-based loop "
                         "would leave the ranks with different numbers of collectives")
    ap.add_argument("--internal-warmup-s", type=float, default=None, help="(older spelling) converted to launches at 4 ms each")
    ap.add_argument("--dist-backend", default="nccl", help="nccl = RCCL over xGMI (the product path); gloo only for the two-ranks-on-one-GPU test")
    ap.add_argument("--share-gpu", action="store_true", help="testing only: every rank uses GPU 0 (needs --dist-backend gloo: RCCL refuses two ranks on one device)")
    ap.add_argument("--samples", type=int, default=30, help="repeats of the K-step region after the timed one (median / p10 / p90)")
    ap.add_argument("--row-stride", type=int, default=384,
                    help="bytes between packed records: 384 = every record as three whole 128-byte lines (the fast layout), 352 = dense")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-small-n", action="store_true", help="skip the 4 096-env sub-record")
    ap.add_argument("--no-step-path", action="store_true", help="skip the bg_step / bg_step_many sample")
    ap.add_argument("--no-gather", action="store_true", help="N > 1: leave the gather of the current observation out")
    ap.add_argument("--gather", choices=["peer", "rccl"], default="peer",
                    help="N > 1: peer = the engine's copy-out writes every env's current record straight into every rank's gather buffer (CUDA IPC mappings over xGMI: "
                         "inside the launch); rccl = an all_gather on a side stream behind the launch.  peer falls back to rccl where the mapping is not available")
    ap.add_argument("--force-gather", action="store_true", help="N = 1: run the gather path anyway (a one-rank RCCL group; exercises the N > 1 code on a one-GPU box)")
    args = ap.parse_args()
    if args.internal_warmup_s is not None:
        args.internal_warmup_launches = int(round(args.internal_warmup_s / 0.004))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus, 1 if args.share_gpu else args.gpus))

    import torch
    import torch.distributed as dist
    from balatro_gym_amd import BalatroVecEnv
    from balatro_gym_amd.sharded import all_gather_bytes, shard_range
    from balatro_gym_amd.vec_env import ObsBuffers, RowBuffers

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = 0 if args.share_gpu else int(os.environ.get("LOCAL_RANK", "0"))
    use_dist = world > 1 or args.force_gather
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        dist.init_process_group(args.dist_backend, rank=rank, world_size=world)
    if args.gpus != world:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: start one rank per GPU (torch.distributed.run --nproc-per-node "
              f"{args.gpus}) or leave WORLD_SIZE unset and let bench.py start the ranks", file=sys.stderr)
        sys.exit(2)
    dev = torch.device(f"cuda:{local_rank}")
    n = args.envs_per_gpu
    total = n * world
    lo, hi = shard_range(total, world, rank)
    assert hi - lo == n

    def make_env(**kw):
        e = BalatroVecEnv(n, [1000 + g for g in range(lo, hi)], device=local_rank, scorer_jokers=True, autoreset=True,
                          max_ante=MAX_ANTE, **kw)
        e.inject(jokers=[jokers_for(g) for g in range(lo, hi)], apply_now=True)
        return e

    env = make_env()
    # chunk = steps per bg_rollout call = what the library fuses into one launch (ring depths: bg_create / BG_KG,KS,KD)
    chunk = args.chunk or int(os.environ.get("BG_BENCH_CHUNK", "0")) or min(372, env.max_fused_steps)
    ob = None
    if args.keep_obs and chunk > 1:
        ob = RowBuffers(n, dev, steps=chunk, row_stride=args.row_stride) if args.obs_layout == "rows" else ObsBuffers(n, dev, steps=chunk)

    # the design's one collective: all_gather of the CURRENT observation record of every env, once per launch, on a side
    # stream so that it runs beside the next launch (what a central evaluator / logger sees; learners train on their own shard)
    do_gather = use_dist and not args.no_gather and isinstance(ob, RowBuffers)
    # the gather WITHOUT a collective (round 5): every rank's engine writes the current record of its envs into every rank's buffer from its copy-out --
    # peer-mapped stores while the launch runs; the barrier that ends the timed region is all the synchronisation it needs
    peer_buf = None
    if do_gather and args.gather == "peer":
        from balatro_gym_amd.sharded import setup_peer_gather
        peer_buf = setup_peer_gather(env, rank, world)
    gather_method = "none" if not do_gather else ("peer" if peer_buf is not None else "rccl")
    do_gather = do_gather and peer_buf is None   # (from here on: the RCCL path)
    gather_stream = torch.cuda.Stream(device=dev) if do_gather else None
    # N > 1: the launch's last record row of every shard is gathered (RCCL all_gather) on a side stream beside the NEXT launch.  Two
    # record buffers alternate, so the next launch never writes what the gather is still reading; a buffer is reused only after the
    # gather that read it (two launches earlier) has completed.  (RCCL's workgroups are multi-wave: they cannot be placed beside
    # a resident step-engine workgroup, so the gather really runs in the tail of the launch it is queued behind.)
    gathered = torch.empty((world, n, 352), dtype=torch.uint8, device=dev) if do_gather else None
    bufs = [ob, RowBuffers(n, dev, steps=chunk, row_stride=args.row_stride)] if do_gather else [ob]
    gather_done = [None] * len(bufs)
    launch_no = 0
    gather_bytes = 0

    def run(nsteps, t0):
        nonlocal gather_bytes, launch_no
        done = 0
        while done < nsteps:
            c = min(chunk, nsteps - done)
            b = launch_no % len(bufs)
            cur = torch.cuda.current_stream(dev)
            if gather_done[b] is not None:
                cur.wait_event(gather_done[b])
            env.rollout(c, policy=POLICY_CYCLE3, policy_seed=POLICY_SEED, env_index0=lo, t0=t0 + done,
                        obs_buffers=bufs[b], zero_stats=False)  # a shorter last call fills the first c rows
            if do_gather:
                gather_stream.wait_stream(cur)
                with torch.cuda.stream(gather_stream):
                    all_gather_bytes(gathered.view(-1), bufs[b].rows[c - 1][:, :352].reshape(-1))
                    ev = torch.cuda.Event()
                    ev.record(gather_stream)
                gather_done[b] = ev
                gather_bytes += n * 352
            launch_no += 1
            done += c
        if do_gather:
            torch.cuda.current_stream(dev).wait_stream(gather_stream)

    def barrier():
        if world > 1:
            dist.barrier()
        # (poll an event first: the synchronize that follows then returns at once instead of after the runtime's wake-up latency)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(dev))
        while not ev.query():
            pass
        torch.cuda.synchronize(dev)

    def stream_barrier():
        """The launch stream only: what a consumer of the records waits for.  The library's side stream (a look-ahead refill that was queued beside a
        launch and serves the NEXT ~18 launches) keeps running, beside whatever is launched next -- where `samples` and `sustained` then see it."""
        if world > 1:
            dist.barrier()
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(dev))
        while not ev.query():
            pass
        ev.synchronize()

    # ---- untimed internal warm-up at full depth (a fixed number of launches: every rank issues the same collectives), then the W
    # warm-up steps of the contract
    t_off = 0
    left = args.internal_warmup_launches
    while left > 0:
        k = min(8, left)
        run(k * chunk, t_off)  # launches back to back, as in the timed region (the refill of one runs beside the next)
        torch.cuda.synchronize(dev)
        t_off += k * chunk
        left -= k
    # (round 4 ran four untimed launches of the timed shape here "so that the lazy refill falls outside" the timed one.  They are gone: the W warm-up
    #  steps of the contract are the only launches in front of the timed region.  The refill -- the MT19937 seeding and the deck shuffles, row a1/a2 work --
    #  is due every 18th launch at 20 steps and issued in PIECES beside the launches that follow (profiles/r05/refill_pieces.txt; a full-depth-sliced
    #  refill beside every short launch was measured too: -21 %, profiles/r05/refill_policy_ab.txt): the timed launch carries a piece, `sustained` a period.)
    gather_verified = None
    if peer_buf is not None and launch_no > 0:
        # The peer-written gather buffers after the warm-up's launches, checked before anything is timed -- BYTE FOR BYTE against one untimed all_gather
        # (RCCL over xGMI; host-staged under gloo) of the same rows: every slot of this rank's buffer, i.e. what every OTHER rank's engine wrote into
        # it through its peer mapping, must be exactly the last record row of that rank's last launch (a slot that was only partly written, or
        # written through a mapping that does not carry non-temporal stores on this box, fails here and not in a learner).  On a mismatch on ANY
        # rank every rank falls back to the RCCL all_gather.
        barrier()
        last = bufs[(launch_no - 1) % len(bufs)].rows[chunk - 1][:, :352].contiguous()
        chk = torch.empty((world, n, 352), dtype=torch.uint8, device=dev)
        all_gather_bytes(chk.view(-1), last.view(-1))
        torch.cuda.synchronize(dev)
        good = bool(torch.equal(peer_buf, chk))
        flag = torch.tensor([1 if good else 0], dtype=torch.int32, device=dev)
        if world > 1:
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        gather_verified = int(flag.item()) == 1
        del chk
        if not gather_verified:
            print("bench.py: peer-written gather buffers differ from the all_gather of the same rows after the warm-up; falling back to the RCCL all_gather", file=sys.stderr)
            from balatro_gym_amd.sharded import teardown_peer_gather
            teardown_peer_gather(env, rank)   # every engine stops writing peers, a barrier, then the buffers the others have mapped may go
            peer_buf, gather_method, do_gather = None, "rccl (peer writes failed their check)", True
            gather_stream = torch.cuda.Stream(device=dev)
            gathered = torch.empty((world, n, 352), dtype=torch.uint8, device=dev)
            bufs = [ob, RowBuffers(n, dev, steps=chunk, row_stride=args.row_stride)]
            gather_done = [None] * len(bufs)
    # (the error check and the zeroing of the statistics come BEFORE the W warm-up steps: nothing but the barrier may stand between the
    #  warm-up and the timed region, or the timed launch starts on a GPU that has idled through a host round trip)
    env.check()
    env._stats.zero_()
    run(args.warmup, t_off)
    t_off += args.warmup
    env.set_profiling(True)
    gather_bytes = 0
    barrier()
    t_start = time.perf_counter()
    run(args.steps, t_off)
    barrier()
    elapsed = time.perf_counter() - t_start
    t_off += args.steps
    prof = env.get_profile()
    stats = env.stats()  # also checks the device error word
    stats["steps"] -= n * args.warmup   # (the statistics were zeroed before the warm-up steps)
    gather_bytes_timed = gather_bytes

    # ---- the same K-step region again, `--samples` times (each between two synchronisations; every rank runs the same count), then
    # a SUSTAINED window: enough back-to-back repeats to span one lazy-refill period, no synchronisation in between
    sample_s = []
    barrier()
    for _ in range(max(0, args.samples)):
        stream_barrier()
        t0s = time.perf_counter()
        run(args.steps, t_off)
        stream_barrier()
        sample_s.append(time.perf_counter() - t0s)
        t_off += args.steps
    barrier()
    prof_samples = env.get_profile()
    reps = max(2, -(-env.max_fused_steps // max(1, args.steps)) + 1)
    barrier()
    t0s = time.perf_counter()
    for _ in range(reps):
        run(args.steps, t_off)
        t_off += args.steps
    barrier()
    sustained_s = time.perf_counter() - t0s
    prof_sustained = env.get_profile()
    env.set_profiling(False)
    env.check()
    if world > 1 and sample_s:
        ts = torch.tensor(sample_s + [sustained_s], dtype=torch.float64, device=dev)
        dist.all_reduce(ts, op=dist.ReduceOp.MAX)
        sample_s, sustained_s = ts[:-1].tolist(), float(ts[-1].item())

    tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    agg = torch.tensor([stats["steps"], stats["episodes"], stats["plays"]], dtype=torch.int64, device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(agg, op=dist.ReduceOp.SUM)
    elapsed = float(tmax.item())
    env_steps = int(agg[0].item())
    assert env_steps == total * args.steps, (env_steps, total, args.steps)

    if rank == 0:
        value = env_steps / elapsed
        # the kernel behind packed-record rollouts: bg_engine3.h (owner waves + service waves) unless BG_ENGINE says otherwise; per-key output: bg_engine.h
        kernel = {"1": "bg_engine_kernel", "2": "bg_owner_kernel", "3": "bg_engine3_kernel"}.get(os.environ.get("BG_ENGINE", "3"), "bg_engine3_kernel") if args.obs_layout == "rows" else "bg_engine_kernel"
        # roofline of the dominant kernel: algorithmic bytes per launch / mean launch duration (HIP events on the launch stream)
        launches = max(1, prof["rollout_launches"])
        fused = prof["rollout_fused_steps"] / launches           # mean fused steps T per launch
        a_step = OBS_BYTES + IO_BYTES + 2 * STATE_BYTES / max(1.0, fused)
        bytes_per_launch = a_step * n * fused
        mean_launch_s = prof["rollout_ms"] / launches * 1e-3
        achieved = bytes_per_launch / mean_launch_s / 1e9 if mean_launch_s > 0 else 0.0
        traffic = matching_traffic(kernel, n, fused) if args.obs_layout == "rows" else None
        peak_measured, peak_write = measured_copy_gbps(dev)
        write_gbps = (OBS_ROW_WRITE_BYTES + STATE_BYTES / max(1.0, fused)) * n * fused / mean_launch_s / 1e9 if mean_launch_s > 0 else 0.0
        out = {
            "metric": "env-steps/sec at 65536 envs, random policy; achieved HBM GB/s vs peak",
            "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "int64/f64", "data": "synthetic",
            "config": {"workload": "BASELINE configs[2]: 65536 envs per GPU, 5 random implemented jokers per env "
                                   "(scorer-level joker chain), Antes 1-4 cap, counter-hash random policy (blind "
                                   "45/46/47 by env index, shop->31, else uniform over valid), SAME_STEP auto-reset, "
                                   "every step's 330-byte observation written to HBM"
                                   + ((" as one packed 352-byte record per (step, env) incl. reward/action/terminated"
                                       + (", records 384 bytes apart and written as three whole 128-byte lines" if args.row_stride == 384 else ""))
                                      if args.obs_layout == "rows" else " as one [T, N] array per key"),
                       "row_stride_bytes": args.row_stride if args.obs_layout == "rows" else None,
                       "obs_layout": args.obs_layout,
                       "envs_per_gpu": n, "total_envs": total, "fused_steps_per_launch": fused,
                       "internal_warmup_launches": args.internal_warmup_launches,
                       "ring_depths": {k: os.environ.get(k, "default") for k in ("BG_KG", "BG_KS", "BG_KD")},
                       "parallelism": f"shard{world} (independent envs, no data-path collective)"},
            # `achieved` / `frac`: algorithmic bytes of the timed region / its WALL time (everything between the two synchronisations: launch
            # latency, the refill beside the launch, the synchronisation itself) -- what the job delivers.  `kernel_achieved` / `kernel_frac`: the
            # same bytes / the dominant kernel's mean launch duration (HIP events on the launch stream; agrees with rocprofv3 --kernel-trace
            # --stats, profiles/) -- what the kernel does while it runs.
            "roofline": {"bound": "hbm", "achieved": value * a_step / world / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": value * a_step / world / 1e9 / HBM_PEAK_GBPS,
                         "frac_is": "wall time of the timed region (per GPU); kernel_frac = HIP-event launch time of the dominant kernel",
                         "kernel_achieved": achieved, "kernel_frac": achieved / HBM_PEAK_GBPS,
                         "traffic": traffic[0] if traffic else None,
                         "traffic_source": traffic[1] if traffic else None,
                         # the PMC counters cannot be read from inside this process: when `traffic` is not null it comes from a committed
                         # rocprofv3 measurement of THIS device code and launch shape on another box
                         "traffic_measured_in_run": False,
                         # the bytes the HBM really moved per second during a launch (PMC traffic / launch time), when known
                         "traffic_gbps": traffic[0] / mean_launch_s / 1e9 if (traffic and mean_launch_s > 0) else None,
                         "traffic_frac_of_measured": traffic[0] / mean_launch_s / 1e9 / peak_measured if (traffic and mean_launch_s > 0 and peak_measured) else None,
                         "peak_measured": peak_measured, "frac_of_measured": value * a_step / world / 1e9 / peak_measured if peak_measured else None,
                         "kernel_frac_of_measured": achieved / peak_measured if peak_measured else None,
                         "peak_guide": HBM_PEAK_GUIDE_GBPS, "frac_of_guide": value * a_step / world / 1e9 / HBM_PEAK_GUIDE_GBPS,
                         # stores only: the records + the state written back, against a plain fill kernel on this GPU
                         "write_gbps": write_gbps, "peak_measured_write": peak_write, "write_frac_of_measured": write_gbps / peak_write if peak_write else None,
                         "kernel": kernel, "algorithmic_bytes_per_env_step": a_step,
                         "mean_launch_us": mean_launch_s * 1e6, "launches": launches,
                         "refill_mean_launch_us": prof["refill_ms"] / max(1, prof["refill_launches"]) * 1e3,
                         "refill_launches_are": "kernel groups of the look-ahead refill issued inside the region: a whole refill beside a 372-step launch, ONE PIECE (the scan, or a dense kernel over a part of a work list) beside a short one",
                         "refill": "the RNG look-ahead refill (deck shuffles, shop-stream seeding, global-stream blocks: 2.4 ms of kernels per 372 env steps) is due whenever the steps launched since the last one would exhaust half a ring -- every launch at 372 fused steps, every 18th at 20 -- and runs on a side stream BESIDE the launches (one SIMD per CU is left to it).  Beside short launches it is issued in pieces, one per launch (BG_REFILL_SLICED), each small enough to be resident beside the engine at once: none is left to be placed in the gap before the next launch, where it would take the registers that launch's workgroups need (that made one launch in 18 2.5x slower)"},
            "episodes": int(agg[1].item()), "accepted_plays": int(agg[2].item()), "episode_counts_cover": "warm-up steps + timed steps",
            "state_bytes_per_gpu": env.state_bytes(),
        }
        if sample_s:
            vs = sorted(total * args.steps / t for t in sample_s)
            pick = lambda q: vs[min(len(vs) - 1, max(0, int(round(q * (len(vs) - 1)))))]
            nl = max(1, prof_samples["rollout_launches"])
            out["samples"] = {"n": len(vs), "what": f"the same {args.steps}-step region again, each between two synchronisations of the LAUNCH stream (the records are complete; the piece of the look-ahead refill beside a launch may end a few microseconds behind it)",
                              "min_over_median": vs[0] / pick(0.5),
                              "median": pick(0.5), "p10": pick(0.1), "p90": pick(0.9), "min": vs[0], "max": vs[-1],
                              "value_inside_p10_p90": bool(pick(0.1) <= value <= pick(0.9)),
                              "mean_launch_us": prof_samples["rollout_ms"] / nl * 1e3, "launches": nl,
                              "refill_launches": prof_samples["refill_launches"],
                              "median_roofline_frac": pick(0.5) * a_step / world / 1e9 / HBM_PEAK_GBPS}
        out["sustained"] = {"value": total * args.steps * reps / sustained_s, "unit": "env-steps/s", "regions": reps,
                            "what": f"{reps} back-to-back repeats of the {args.steps}-step region, no synchronisation in between",
                            "refill_launches_inside": prof_sustained["refill_launches"], "launches": prof_sustained["rollout_launches"],
                            "over_value": total * args.steps * reps / sustained_s / value,
                            "roofline_frac": total * args.steps * reps / sustained_s * a_step / world / 1e9 / HBM_PEAK_GBPS}
        out["roofline"]["refill_launches_in_timed_region"] = prof["refill_launches"]
        out["roofline"]["refill_kernel_us_in_timed_region"] = prof["refill_ms"] * 1e3
        if use_dist:
            out["gather"] = {"in_timed_region": gather_method != "none", "method": gather_method,
                             "what": ("the engine's copy-out writes the current 352-byte record of every env into every rank's gather buffer (peer-mapped stores over xGMI, inside the launch); "
                                      "the region's closing barrier is its synchronisation") if gather_method == "peer" else
                                     "all_gather_into_tensor of the current 352-byte record of every env, once per launch, side stream",
                             "bytes_per_gpu_per_launch": n * 352 if gather_method == "peer" else ((gather_bytes_timed // max(1, launches)) if do_gather else 0),
                             # peer writes: every slot of this rank's buffer == an untimed all_gather of the same rows, byte for byte, on every rank (checked after the warm-up)
                             "verified": gather_verified}
    env.close()
    del ob, bufs

    if rank == 0 and world == 1 and not args.no_small_n:
        # ---- BASELINE configs[1]'s size: 4 096 envs on one GPU (what the reference's own users run is smaller still), same workload and launch shape
        sn = 4096
        es = BalatroVecEnv(sn, [1000 + g for g in range(sn)], device=local_rank, scorer_jokers=True, autoreset=True, max_ante=MAX_ANTE)
        es.inject(jokers=[jokers_for(g) for g in range(sn)], apply_now=True)
        cs = min(372, es.max_fused_steps)
        rbs = RowBuffers(sn, dev, steps=cs, row_stride=args.row_stride)
        for i in range(4):
            es.rollout(cs, policy=POLICY_CYCLE3, policy_seed=POLICY_SEED, t0=i * cs, obs_buffers=rbs, zero_stats=False)
        torch.cuda.synchronize(dev)
        es.set_profiling(True)
        t0s = time.perf_counter()
        for i in range(8):
            es.rollout(cs, policy=POLICY_CYCLE3, policy_seed=POLICY_SEED, t0=(4 + i) * cs, obs_buffers=rbs, zero_stats=False)
        torch.cuda.synchronize(dev)
        dts = time.perf_counter() - t0s
        ps = es.get_profile()
        es.check()
        es.close()
        a_s = OBS_BYTES + IO_BYTES + 2 * STATE_BYTES / cs
        out["small_n"] = {"envs": sn, "value": sn * cs * 8 / dts, "unit": "env-steps/s", "fused_steps_per_launch": cs,
                          "mean_launch_us": ps["rollout_ms"] / max(1, ps["rollout_launches"]) * 1e3,
                          "roofline_frac": sn * cs * a_s / (ps["rollout_ms"] / max(1, ps["rollout_launches"]) * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                          "what": "BASELINE configs[1]'s env count with the workload of the headline line: 8 launches back to back"}
        del rbs

    if rank == 0 and world == 1 and not args.no_step_path:
        # ---- the Gymnasium-surface path: bg_step with actions from a device tensor (recorded from a fused rollout of a twin
        # handle, so every action is the valid policy action of that state), then the same steps through bg_step_many
        ks = 800   # more than two refill periods (372 steps): the look-ahead refill runs beside these launches too (in pieces), and is inside the figure
        twin = make_env()
        acts = torch.zeros((ks, n), dtype=torch.int32, device=dev)
        # (recorded in pieces WITH per-step observation buffers: without them every step of a rollout writes row 0 of its outputs -- until round 6 this
        #  block replayed rows of zeros, i.e. 800 rejected PLAY_HANDs per env: figures of a step path that never ran a service step)
        rec = 50
        rec_ob = ObsBuffers(n, dev, steps=rec)
        for c0 in range(0, ks, rec):
            twin.rollout(rec, policy=POLICY_CYCLE3, policy_seed=POLICY_SEED, env_index0=lo, t0=c0, obs_buffers=rec_ob, actions=acts[c0:c0 + rec], zero_stats=c0 == 0)
        twin_stats = twin.stats()
        twin.close()
        del rec_ob
        replay_plays = int((acts == 0).sum().item())   # (PLAY_HAND is action 0: the policy only plays it when it is valid)
        res = {}
        kb = 100   # bg_step_many_kept: steps per call, every step's observation kept in [kb, N] buffers (2.3 GB)
        kept_ob = ObsBuffers(n, dev, steps=kb)
        kept_rw = torch.zeros((kb, n), dtype=torch.float64, device=dev)
        kept_tm = torch.zeros((kb, n), dtype=torch.uint8, device=dev)
        for mode in ("bg_step", "bg_step_rows", "bg_step_many", "bg_step_many_kept"):
            kw = {"obs_layout": "rows"} if mode == "bg_step_rows" else {}   # bg_step_rows: the observation as one packed 384-byte record per env
            e2 = make_env(**kw)
            e2.step(acts[0]); e2.reset(); torch.cuda.synchronize(dev)  # first-call costs out of the way
            e2.close()
            # two passes over the same 200 steps: the wall clock WITHOUT the profiling events (two hipEventRecord per launch cost a
            # one-step launch a fifth of its time), then the kernel time WITH them
            dt, p = None, None
            for profiled in (False, True):
                e2 = make_env(**kw)
                e2.set_profiling(profiled)
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                if mode == "bg_step_many":
                    e2.step_many(acts)   # the observation arrays are overwritten in place: the caller sees the LAST step's (the library writes them once per launch)
                elif mode == "bg_step_many_kept":
                    for c0 in range(0, ks, kb):
                        e2.step_many(acts[c0:c0 + kb], obs_buffers=kept_ob, reward=kept_rw, terminated=kept_tm)
                else:
                    for k in range(ks):
                        e2.step(acts[k])
                torch.cuda.synchronize(dev)
                if profiled:
                    p = e2.get_profile()
                else:
                    dt = time.perf_counter() - t0
                e2.check()
                e2.close()
            a1 = OBS_BYTES + IO_BYTES + 2 * STATE_BYTES  # one launch per step: the state crosses HBM every step
            a_k = OBS_BYTES + IO_BYTES + 2 * STATE_BYTES / ks                 # every step's observation kept
            a_last = IO_BYTES + (OBS_BYTES + 2 * STATE_BYTES) / ks             # only the last step's observation leaves the chip
            alg = {"bg_step_many": a_last, "bg_step_many_kept": a_k}.get(mode, a1) * n * ks
            res[mode] = {"value": n * ks / dt, "unit": "env-steps/s", "steps": ks, "ms_per_step": dt / ks * 1e3,
                         "kernel_ms_per_step": p["step_ms"] / ks, "launches": p["step_launches"],
                         "wall_over_kernel": (dt / ks * 1e3) / (p["step_ms"] / ks) if p["step_ms"] > 0 else None,
                         "roofline_frac": alg / (p["step_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBPS if p["step_ms"] > 0 else None}
        del kept_ob, kept_rw, kept_tm
        out["step_path"] = {"what": f"{ks} steps of the same workload, actions from a device tensor [K, N]; bg_step / bg_step_many*: observation as one array per key, "
                                    "bg_step_rows: as one packed 384-byte record per env (every key a strided view); bg_step_many: ONE call, the observation arrays "
                                    f"overwritten in place (the last step's is what the caller sees); bg_step_many_kept: {kb} steps per call, every step's observation, reward and "
                                    "termination flag kept in [K, N] buffers",
                            "twin_rollout_plays": twin_stats["plays"], "replayed_play_actions": replay_plays,
                            "note": "until round 6 this block replayed rows of ZERO actions (rejected PLAY_HANDs, no service step): its earlier figures (3.2 / 3.7 / 7.9 G) are not comparable",
                            **res}

    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            threads = os.cpu_count() or 1
            n_cpu, t_cpu = 256 * threads, 3000
            v, dt = cpu_baseline(n_cpu, t_cpu, threads)
            if dt < 8.0:  # scale the sample to ~10-30 s of CPU work
                t_cpu = int(t_cpu * 15.0 / max(dt, 1e-3))
                v, dt = cpu_baseline(n_cpu, t_cpu, threads)
            out["cpu_baseline"] = {"value": v, "unit": "env-steps/s", "cores": threads, "kind": "port",
                                   "sample": f"{n_cpu} envs x {t_cpu} steps of the same workload on the C oracle "
                                             f"(oracle/balatro_oracle.c), {threads} threads, {dt:.1f} s",
                                   # The reference itself (pure Python) cannot travel to this box.  In the build container the C oracle runs the
                                   # reference's own smoke workload 390x (1 process) / 399x (8 processes) faster than balatro_env_2.BalatroEnv
                                   # (4.86 M vs 12.5 k, 43.6 M vs 109 k env-steps/s): an ESTIMATE of what the Python
                                   # reference would do on these host cores, one process per thread
                                   "python_reference_estimate": {"value": v / 390.0, "unit": "env-steps/s", "cores": threads,
                                                                 "how": "cpu_baseline.value / 390 (oracle : reference ratio measured in the build container, BASELINE.md section 4); an estimate, not a measurement on this box"}}
        # RCCL prints a version banner through C stdio on rank 0; flush it first so that the JSON line is the LAST line of stdout
        try:
            C.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.flush()
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
