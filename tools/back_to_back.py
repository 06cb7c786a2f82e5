#!/usr/bin/env python3
"""Development aid: K fused launches of T steps issued back to back (no synchronisation in between) -- wall time per launch against the kernel's own duration,
i.e. what the stream loses between two launches (dependencies, events, the lazy refill)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from balatro_gym_amd import BalatroVecEnv
from balatro_gym_amd.vec_env import RowBuffers
from bench import jokers_for, POLICY_CYCLE3, POLICY_SEED
n, T = 65536, int(os.environ.get("T", "20"))
env = BalatroVecEnv(n, [1000 + g for g in range(n)], device=0, scorer_jokers=True, autoreset=True, max_ante=4)
env.inject(jokers=[jokers_for(g) for g in range(n)], apply_now=True)
rb = RowBuffers(n, env.device, steps=T, row_stride=384)
t0 = 0
for i in range(60):
    env.rollout(T, policy=POLICY_CYCLE3, policy_seed=POLICY_SEED, t0=t0, obs_buffers=rb, zero_stats=False); t0 += T
torch.cuda.synchronize()
for K in (1, 4, 8, 16, 32, 64):
    best = None
    for rep in range(5):
        torch.cuda.synchronize(); a = time.perf_counter()
        for i in range(K):
            env.rollout(T, policy=POLICY_CYCLE3, policy_seed=POLICY_SEED, t0=t0, obs_buffers=rb, zero_stats=False); t0 += T
        torch.cuda.synchronize(); b = time.perf_counter()
        w = (b - a) * 1e6 / K
        best = w if best is None or w < best else best
    print(f"K {K:3d} launches of {T} steps back to back: best {best:7.1f} us per launch -> {n * T / best / 1e3:5.2f} G env-steps/s", flush=True)
env.set_profiling(True)
env.rollout(T, policy=POLICY_CYCLE3, policy_seed=POLICY_SEED, t0=t0, obs_buffers=rb, zero_stats=False); torch.cuda.synchronize()
print("kernel alone:", round(env.get_profile()["rollout_ms"] * 1e3, 1), "us")
env.close()
