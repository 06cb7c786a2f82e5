#!/usr/bin/env python3
"""Ablation timings of the fused rollout kernel on one GPU (development tool, not part of the product)."""
import ctypes as C
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from balatro_gym_amd import BalatroVecEnv  # noqa: E402
from balatro_gym_amd import _native as nat  # noqa: E402
from balatro_gym_amd.vec_env import ObsBuffers, RowBuffers  # noqa: E402
from bench import IMPLEMENTED, jokers_for  # noqa: E402


def run(n, scorer, policy, chunk, steps, obs_mode, max_ante=4, label=""):
    env = BalatroVecEnv(n, [1000 + i for i in range(n)], device=0, scorer_jokers=scorer, autoreset=True, max_ante=max_ante)
    if scorer:
        env.inject(jokers=[jokers_for(i) for i in range(n)], apply_now=True)
    ob = None
    if obs_mode == "keep":
        ob = ObsBuffers(n, env.device, steps=chunk)
    elif obs_mode == "rows":
        ob = RowBuffers(n, env.device, steps=chunk)
    elif obs_mode == "none":
        ob = ObsBuffers(1, env.device)
        ob.ptrs = nat.ObsPtrs()  # all NULL
        ob.steps = 1
    def go(k, t0):
        done = 0
        while done < k:
            env.rollout(chunk, policy=policy, policy_seed=7, t0=t0 + done, obs_buffers=ob, zero_stats=False)
            done += chunk
    go(chunk * 2, 0)
    env.check()
    env.set_profiling(True)
    torch.cuda.synchronize()
    t = time.perf_counter()
    go(steps, chunk * 2)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    p = env.get_profile()
    env.check()
    fused = p["rollout_fused_steps"]
    print(f"{label:34s} n={n:7d} chunk={chunk:3d} obs={obs_mode:5s} | rollout {p['rollout_ms']*1e3/fused:7.2f} us/step "
          f"| refill {p['refill_ms']*1e3/fused:7.2f} us/step | wall {dt*1e6/steps:7.2f} us/step | {n*steps/dt/1e6:8.1f} M steps/s",
          flush=True)
    if os.environ.get("BG_TIMING"):
        import ctypes as C2
        out = (C2.c_ulonglong * 32)()
        L = nat.load()
        L.bg_debug_counters.argtypes = [C2.c_void_p, C2.POINTER(C2.c_ulonglong)]
        L.bg_debug_counters(env._h, out)
        blocks = (n + 127) // 128  # BG_RB
        T = max(1, out[4] // blocks)
        print(f"    phase cycles per block-step: A {out[0]/blocks/T:9.0f}  B {out[1]/blocks/T:9.0f}  C {out[2]/blocks/T:9.0f}  items/block-step {out[3]/blocks/T:6.1f}")
        iters, rounds = out[15] & 0xffffffff, out[15] >> 32
        print(f"    per block-step: iterations {iters/blocks/T:5.2f}  phase-B rounds {rounds/blocks/T:5.2f}  items per round {out[3]/max(1,rounds):5.1f}"
              f"  | cycles per iteration A {out[0]/max(1,iters):7.0f} C {out[2]/max(1,iters):7.0f}  per round B {out[1]/max(1,rounds):7.0f}")
        print("    A/C sections, cycles per wave-iteration: " + " | ".join(f"{nm} {out[i]/max(1,iters)/4:.0f}" for i, nm in
              [(20, "policy"), (21, "guards+cheap+enqueue"), (16, "merge"), (17, "cap+reset"), (18, "mask"), (19, "obs")]))
        if out[26]:
            print(f"    iterations per workgroup-launch: mean {iters/max(1,out[26]):.1f}  max {out[24]}  min {(~out[25]) & 0xffffffffffffffff} (max/min are over all launches)")
        print(f"    phase-B wave time per round: plays {out[22]/max(1,rounds):.0f}  others {out[23]/max(1,rounds):.0f}")
        names = {5: "gather", 6: "classify", 7: "boss-check+joker-individual", 8: "bloodstone+skip", 9: "joker-main", 10: "boss-ratio+state", 11: "reward", 12: "outcome", 13: "main-prefetch", 14: "main-loop"}
        print("    play path cycles per block-step: " + " | ".join(f"{names[i]} {out[i]/blocks/T:.0f}" for i in range(5, 15)))
    env.close()


if __name__ == "__main__":
    chunk = int(os.environ.get("CHUNK", "16"))
    steps = chunk * 8
    run(65536, True, 2, chunk, steps, "rows", label="C3 packed records (bench)")
    run(65536, True, 2, chunk, steps, "keep", label="C3 one array per key")
    if os.environ.get("BG_TIMING"):
        run(65536, False, 2, chunk, steps, "keep", max_ante=0, label="C2 no jokers")
        sys.exit(0)
    run(65536, True, 2, chunk, steps, "live", label="C3 obs overwritten in place")
    run(65536, True, 2, chunk, steps, "none", label="C3 no obs writes")
    run(65536, False, 2, chunk, steps, "keep", max_ante=0, label="C2 no jokers")
    run(65536, False, 1, chunk, steps, "keep", max_ante=0, label="C1 small-only policy")
    run(65536, False, 0, chunk, steps, "keep", max_ante=0, label="C5 uniform policy")
    run(131072, True, 2, chunk, steps, "keep", label="C3 2x envs")
    run(262144, True, 2, chunk, steps, "keep", label="C3 4x envs")
    run(16384, True, 2, chunk, steps, "keep", label="C3 1/4 envs")
