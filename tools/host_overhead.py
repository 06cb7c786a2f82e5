#!/usr/bin/env python3
"""Development: where the wall time of ONE 20-step region goes beyond its kernel (bench.py's contract region at the driver's shape): the rollout call's
host time, the wait for an event recorded behind the launch, torch.cuda.synchronize(); with and without the per-launch profile."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from balatro_gym_amd import BalatroVecEnv
from balatro_gym_amd.vec_env import RowBuffers
from bench import jokers_for, POLICY_CYCLE3, POLICY_SEED
n = 65536
env = BalatroVecEnv(n, [1000 + g for g in range(n)], device=0, scorer_jokers=True, autoreset=True, max_ante=4)
env.inject(jokers=[jokers_for(g) for g in range(n)], apply_now=True)
rb = RowBuffers(n, env.device, steps=20, row_stride=384)
for i in range(60):
    env.rollout(20, policy=POLICY_CYCLE3, policy_seed=POLICY_SEED, t0=i * 20, obs_buffers=rb, zero_stats=False)
torch.cuda.synchronize()
med = lambda v: sorted(v)[len(v) // 2]
step = 60
for prof in (False, True):
    env.set_profiling(prof)
    calls, evs, walls, kern = [], [], [], []
    for i in range(60):
        ev = torch.cuda.Event(); ev.record()
        while not ev.query(): pass
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        env.rollout(20, policy=POLICY_CYCLE3, policy_seed=POLICY_SEED, t0=step * 20, obs_buffers=rb, zero_stats=False)
        step += 1
        t1 = time.perf_counter()
        ev = torch.cuda.Event(); ev.record()
        while not ev.query(): pass
        t2 = time.perf_counter()
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        calls.append((t1 - t0) * 1e6); evs.append((t2 - t0) * 1e6); walls.append((t3 - t0) * 1e6)
        if prof:
            p = env.get_profile(); kern.append(p["rollout_ms"] * 1e3 / max(1, p["rollout_launches"]))
    print(f"profiling {prof}: rollout() returns after {med(calls):.1f} us; event behind the launch seen after {med(evs):.1f}; after synchronize {med(walls):.1f} (min {min(walls):.1f})"
          + (f"; kernel (its own timestamps) {med(kern):.1f}" if kern else ""))
env.close()
