#!/usr/bin/env python3
"""BUILD-CONTAINER ONLY (the reference cannot travel): throughput of the Python reference env and of the C oracle on the same
workload, and their ratio -- so that the C-oracle figure measured on the GPU box (`bench.py` cpu_baseline) can be turned into a
clearly labelled Python-reference-equivalent estimate.  BASELINE.md section 3.

Workload C1 of SURVEY 8(d): seed 42, policy `small_only` (balatro_env_2.py:1841-1849: BLIND_SELECT -> 45, SHOP -> 31, else uniform
over valid actions), counter-hash choice so that both sides take the same actions; episodes restart on termination.
usage: python tools/bench_reference.py [steps] [processes]
"""
import multiprocessing as mp
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run_reference(args):
    seed, steps = args
    from oracle import refharness as rh
    env = rh.RefEnv(seed)
    obs = env.obs()
    t0 = time.perf_counter()
    for t in range(steps):
        a = rh.policy_action(obs["action_mask"], int(obs["phase"]), rh.POLICY_SMALL_ONLY, 7, 0, t)
        obs, r, term, _, _ = env.step(a)
        if term:
            obs = env.reset()
    return steps / (time.perf_counter() - t0)


def run_oracle(args):
    seed, steps = args
    from oracle import pyoracle as po
    import ctypes as C
    L = po.lib()
    e = po.OracleEnv(seed)
    arr = (C.c_void_p * 1)(e.handle)
    rs, ss, ep = C.c_double(), C.c_int64(), C.c_int64()
    t0 = time.perf_counter()
    n = L.bo_rollout(arr, 1, 0, steps, po.POLICY_SMALL_ONLY, 7, 0, C.byref(rs), C.byref(ss), C.byref(ep))
    return n / (time.perf_counter() - t0)


if __name__ == "__main__":
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
    procs = int(sys.argv[2]) if len(sys.argv) > 2 else (os.cpu_count() or 1)
    ref1 = run_reference((42, steps))
    orc1 = run_oracle((42, steps * 20))
    with mp.get_context("spawn").Pool(procs) as pool:
        refn = sum(pool.map(run_reference, [(42 + i, steps) for i in range(procs)]))
        orcn = sum(pool.map(run_oracle, [(42 + i, steps * 20) for i in range(procs)]))
    print(f"python reference : 1 process {ref1:10.0f} steps/s   {procs} processes {refn:12.0f} steps/s")
    print(f"C oracle         : 1 thread  {orc1:10.0f} steps/s   {procs} processes {orcn:12.0f} steps/s")
    print(f"ratio oracle / reference: {orc1 / ref1:.0f}x (1 core), {orcn / refn:.0f}x ({procs} cores)")
