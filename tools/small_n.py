#!/usr/bin/env python3
"""Development: env-steps/s of fused packed-record rollouts at small env counts (N=4096 ... via env var NS), configs[2] workload."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from balatro_gym_amd import BalatroVecEnv
from balatro_gym_amd.vec_env import RowBuffers
from bench import jokers_for, POLICY_CYCLE3, POLICY_SEED
for n in [int(v) for v in os.environ.get("NS", "4096").split(",")]:
    env = BalatroVecEnv(n, [1000 + g for g in range(n)], device=0, scorer_jokers=True, autoreset=True, max_ante=4)
    env.inject(jokers=[jokers_for(g) for g in range(n)], apply_now=True)
    T = min(372, env.max_fused_steps)
    rb = RowBuffers(n, env.device, steps=T, row_stride=384)
    for i in range(3):
        env.rollout(T, policy=POLICY_CYCLE3, policy_seed=POLICY_SEED, t0=i * T, obs_buffers=rb, zero_stats=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(8):
        env.rollout(T, policy=POLICY_CYCLE3, policy_seed=POLICY_SEED, t0=(3 + i) * T, obs_buffers=rb, zero_stats=False)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    env.check()
    print(f"N={n} T={T} cfg={os.environ.get('BG_E3_CFG', 'default')} engine={os.environ.get('BG_ENGINE', '3')}: {n * T * 8 / dt / 1e9:.3f} G env-steps/s ({dt / 8 * 1e6:.0f} us per launch)")
    env.close()
