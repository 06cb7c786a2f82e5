#!/usr/bin/env python3
"""Development aid: duration of ONE fused rollout launch of T steps (65 536 envs, the bench workload, packed 384-byte records) for small T --
the fixed part of a launch (prologue: state -> LDS images; epilogue) against its per-step part."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from balatro_gym_amd import BalatroVecEnv
from balatro_gym_amd.vec_env import RowBuffers
from bench import jokers_for, POLICY_CYCLE3, POLICY_SEED
n = int(os.environ.get("N", "65536"))
env = BalatroVecEnv(n, [1000 + g for g in range(n)], device=0, scorer_jokers=True, autoreset=True, max_ante=4)
env.inject(jokers=[jokers_for(g) for g in range(n)], apply_now=True)
rb = RowBuffers(n, env.device, steps=372, row_stride=384)
t0 = 0
for i in range(3):
    env.rollout(372, policy=POLICY_CYCLE3, policy_seed=POLICY_SEED, t0=t0, obs_buffers=rb, zero_stats=False); t0 += 372
torch.cuda.synchronize()
env.set_profiling(True)
for T in (1, 2, 3, 5, 10, 20, 40):
    ts = []
    for rep in range(12):
        env.rollout(T, policy=POLICY_CYCLE3, policy_seed=POLICY_SEED, t0=t0, obs_buffers=rb, zero_stats=False); t0 += T
        torch.cuda.synchronize()
        ts.append(env.get_profile()["rollout_ms"] * 1e3)
    ts.sort()
    print(f"T {T:3d}: launch median {ts[len(ts) // 2]:7.1f} us  min {ts[0]:7.1f}  -> {n * T / ts[len(ts) // 2] / 1e3:6.2f} G env-steps/s", flush=True)
env.close()
