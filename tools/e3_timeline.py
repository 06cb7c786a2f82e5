#!/usr/bin/env python3
"""Development aid: where a SHORT engine-3 launch spends its time -- mean over workgroups of wall-clock stamps (100 MHz) taken at the end of the prologue,
of the owner loops, of the service loops and of the workgroup (build: tools/build_variant.sh e3tl -DBG_E3_TL; nothing else is instrumented)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from balatro_gym_amd import BalatroVecEnv, _native as nat
from balatro_gym_amd.vec_env import RowBuffers
from bench import jokers_for, POLICY_CYCLE3, POLICY_SEED
n = int(os.environ.get("N", "65536"))
env = BalatroVecEnv(n, [1000 + g for g in range(n)], device=0, scorer_jokers=True, autoreset=True, max_ante=4)
env.inject(jokers=[jokers_for(g) for g in range(n)], apply_now=True)
rb = RowBuffers(n, env.device, steps=372, row_stride=384)
t0 = 0
for i in range(3):
    env.rollout(372, policy=POLICY_CYCLE3, policy_seed=POLICY_SEED, t0=t0, obs_buffers=rb, zero_stats=False); t0 += 372
torch.cuda.synchronize()
L = nat.load()
L.bg_debug_counters.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
env.set_profiling(True)
wgs = (n + 255) // 256
for T in [int(x) for x in os.environ.get("TS", "1,2,5,10,20,40").split(",")]:
    a = (C.c_ulonglong * 32)(); b = (C.c_ulonglong * 32)()
    reps = 8
    L.bg_debug_counters(env._h, a)
    us = []
    for rep in range(reps):
        env.rollout(T, policy=POLICY_CYCLE3, policy_seed=POLICY_SEED, t0=t0, obs_buffers=rb, zero_stats=False); t0 += T
        torch.cuda.synchronize(); us.append(env.get_profile()["rollout_ms"] * 1e3)
    L.bg_debug_counters(env._h, b)
    o = [float(b[i]) / reps for i in range(32)]   # (bg_debug_counters reads AND clears)
    us.sort()
    print(f"T {T:3d}: launch {us[len(us) // 2]:6.1f} us | mean over workgroups, us after the workgroup's first instruction: prologue done {o[28] / wgs / 100:5.1f}, "
          f"owner loops end {o[29] / (4 * wgs) / 100:6.1f}, service loops end {o[30] / (3 * wgs) / 100:6.1f}, workgroup end {o[31] / wgs / 100:6.1f}", flush=True)
env.close()
