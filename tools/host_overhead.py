import os, sys, time
sys.path.insert(0, '.')
import torch
from balatro_gym_amd import BalatroVecEnv
from balatro_gym_amd.vec_env import RowBuffers
from bench import jokers_for, POLICY_CYCLE3, POLICY_SEED
n = 65536
env = BalatroVecEnv(n, [1000 + g for g in range(n)], device=0, scorer_jokers=True, autoreset=True, max_ante=4)
env.inject(jokers=[jokers_for(g) for g in range(n)], apply_now=True)
rb = RowBuffers(n, env.device, steps=20, row_stride=384)
for i in range(50):
    env.rollout(20, policy=POLICY_CYCLE3, policy_seed=POLICY_SEED, t0=i * 20, obs_buffers=rb, zero_stats=False)
torch.cuda.synchronize()
calls, walls = [], []
for i in range(40):
    ev = torch.cuda.Event(); ev.record()
    while not ev.query(): pass
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    env.rollout(20, policy=POLICY_CYCLE3, policy_seed=POLICY_SEED, t0=(50 + i) * 20, obs_buffers=rb, zero_stats=False)
    t1 = time.perf_counter()
    ev = torch.cuda.Event(); ev.record()
    while not ev.query(): pass
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    calls.append((t1 - t0) * 1e6); walls.append((t2 - t0) * 1e6)
calls.sort(); walls.sort()
print(f"rollout() call returns after {calls[len(calls)//2]:.1f} us (median); launch + wait {walls[len(walls)//2]:.1f} us; min wall {walls[0]:.1f}")
env.set_profiling(True)
env.rollout(20, policy=POLICY_CYCLE3, policy_seed=POLICY_SEED, t0=5000, obs_buffers=rb, zero_stats=False)
torch.cuda.synchronize()
print(env.get_profile())
