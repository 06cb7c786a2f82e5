#!/usr/bin/env python3
"""Development aid: steady-state refill work-list lengths for the bench workload (sync refill)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["BG_ASYNC_REFILL"] = "0"
import torch
from balatro_gym_amd import BalatroVecEnv, _native as nat
from bench import jokers_for
n = 65536
chunk = int(os.environ.get("CHUNK", "16"))
env = BalatroVecEnv(n, [1000 + i for i in range(n)], device=0, scorer_jokers=True, autoreset=True, max_ante=4)
env.inject(jokers=[jokers_for(i) for i in range(n)], apply_now=True)
L = nat.load()
L.bg_debug_worklists.argtypes = [C.c_void_p, C.POINTER(C.c_uint)]
out = (C.c_uint * 4)()
t0 = 0
for k in range(8):
    env.rollout(chunk, policy=2, policy_seed=7, t0=t0, zero_stats=False); t0 += chunk
    L.bg_debug_worklists(env._h, out)
    print(f"chunk {k}: deck envs {out[0]}  seedring envs {out[1]}  gblk envs {out[2]}  shop streams {out[3]}")
print(env.stats())
