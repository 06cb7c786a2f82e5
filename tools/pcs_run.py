#!/usr/bin/env python3
"""Development: a handful of fused rollouts of the benchmark workload, for `rocprofv3 --pc-sampling-beta-enabled` (T / LAUNCHES / N from the environment)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from balatro_gym_amd import BalatroVecEnv
from balatro_gym_amd.vec_env import RowBuffers
from bench import jokers_for, POLICY_CYCLE3, POLICY_SEED
n, T, L = int(os.environ.get("N", "65536")), int(os.environ.get("T", "372")), int(os.environ.get("LAUNCHES", "4"))
env = BalatroVecEnv(n, [1000 + g for g in range(n)], device=0, scorer_jokers=True, autoreset=True, max_ante=4)
env.inject(jokers=[jokers_for(g) for g in range(n)], apply_now=True)
rb = RowBuffers(n, env.device, steps=T, row_stride=384)
for i in range(L):
    env.rollout(T, policy=POLICY_CYCLE3, policy_seed=POLICY_SEED, t0=i * T, obs_buffers=rb, zero_stats=False)
torch.cuda.synchronize()
env.check()
print("done", env.stats())
env.close()
