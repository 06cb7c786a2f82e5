#!/usr/bin/env python3
"""Development aid: many 372-step record rollouts with the bench workload, one line per slow launch; stops at the first device error."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from balatro_gym_amd import BalatroVecEnv
from balatro_gym_amd.vec_env import RowBuffers
from bench import jokers_for, POLICY_CYCLE3, POLICY_SEED
n, T, L = int(os.environ.get("N", "65536")), int(os.environ.get("T", "372")), int(os.environ.get("L", "60"))
t0 = time.time()
env = BalatroVecEnv(n, [1000 + g for g in range(n)], device=0, scorer_jokers=True, autoreset=True, max_ante=4)
env.inject(jokers=[jokers_for(g) for g in range(n)], apply_now=True)
rb = RowBuffers(n, env.device, steps=T, row_stride=int(os.environ.get("STRIDE", "384")))
print(f"setup {time.time()-t0:.1f} s", flush=True)
for i in range(L):
    t1 = time.time()
    env.rollout(T, policy=POLICY_CYCLE3, policy_seed=POLICY_SEED, t0=i * T, obs_buffers=rb, zero_stats=False)
    torch.cuda.synchronize()
    dt = time.time() - t1
    if dt > 0.02 or i % 20 == 0:
        print(f"launch {i}: {dt*1e3:.1f} ms", flush=True)
    try:
        env.check()
    except Exception as ex:
        print(f"launch {i}: device error: {ex}", flush=True)
        break
print(f"done {time.time()-t0:.1f} s", flush=True)
