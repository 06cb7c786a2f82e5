#!/usr/bin/env python3
"""Development aid: wall time per 64-step launch with individual refill kernels switched off (BG_DEV_SKIP_REFILL bit mask:
1 shop, 2 deck, 4 seed ring, 8 global blocks).  The rings run dry, so only the first few launches are meaningful."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from balatro_gym_amd import BalatroVecEnv
from balatro_gym_amd.vec_env import RowBuffers
from bench import jokers_for
n = 65536
env = BalatroVecEnv(n, [1000 + i for i in range(n)], device=0, scorer_jokers=True, autoreset=True, max_ante=4)
env.inject(jokers=[jokers_for(i) for i in range(n)], apply_now=True)
T = env.max_fused_steps
rb = RowBuffers(n, env.device, steps=T)
t0 = 0
for k in range(6):
    torch.cuda.synchronize(); a = time.perf_counter()
    for _ in range(2):
        env.rollout(T, policy=2, policy_seed=7, t0=t0, obs_buffers=rb, zero_stats=False); t0 += T
    torch.cuda.synchronize(); b = time.perf_counter()
    print(f"skip={os.environ.get('BG_DEV_SKIP_REFILL','0')} pair {k}: {(b-a)*1e6/2:.0f} us per launch")
