#!/usr/bin/env python3
"""Development aid: run a few 372-step launches with the refill kernels selected by BG_DEV_SKIP_REFILL (1 shop, 2 decks, 4 seed ring,
8 global blocks are SKIPPED) and synchronous refill, under `rocprofv3 --kernel-trace --stats`, to time each refill kernel alone on
the GPU.  Skipping a kernel starves its ring (the device error word is ignored here): timing only, the results are garbage."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("BG_ASYNC_REFILL", "0")
import torch
from balatro_gym_amd import BalatroVecEnv
from balatro_gym_amd.vec_env import RowBuffers
from bench import jokers_for, POLICY_CYCLE3, POLICY_SEED
n, T = 65536, 372
skip = os.environ.pop("SKIP", "0")
env = BalatroVecEnv(n, [1000 + g for g in range(n)], device=0, scorer_jokers=True, autoreset=True, max_ante=4)
env.inject(jokers=[jokers_for(g) for g in range(n)], apply_now=True)
rb = RowBuffers(n, env.device, steps=T)
for i in range(3):   # steady state first, with every kernel
    env.rollout(T, policy=POLICY_CYCLE3, policy_seed=POLICY_SEED, t0=i * T, obs_buffers=rb, zero_stats=False)
torch.cuda.synchronize()
import ctypes as C
L = env._L
L.bg_debug_set_skip.argtypes = [C.c_void_p, C.c_int]
L.bg_debug_set_skip(env._h, int(skip))
for i in range(3, 6):
    env.rollout(T, policy=POLICY_CYCLE3, policy_seed=POLICY_SEED, t0=i * T, obs_buffers=rb, zero_stats=False)
torch.cuda.synchronize()
print("done skip", skip, flush=True)
