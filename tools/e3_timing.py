#!/usr/bin/env python3
"""Development: cycle counters of engine 3 (build with tools/build_variant.sh e3t -DBG_E3_TIMING, then
BALATRO_MI355X_LIB=build/variants/e3t.so python tools/e3_timing.py)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from balatro_gym_amd import BalatroVecEnv, _native as nat
from balatro_gym_amd.vec_env import RowBuffers
from bench import jokers_for, POLICY_CYCLE3, POLICY_SEED
n, T = int(os.environ.get("N", "65536")), int(os.environ.get("T", "372"))
env = BalatroVecEnv(n, [1000 + g for g in range(n)], device=0, scorer_jokers=True, autoreset=True, max_ante=4)
env.inject(jokers=[jokers_for(g) for g in range(n)], apply_now=True)
WARM = int(os.environ.get("WARM", "372"))
rb = RowBuffers(n, env.device, steps=max(T, WARM), row_stride=384)
for i in range(3):
    env.rollout(WARM, policy=POLICY_CYCLE3, policy_seed=POLICY_SEED, t0=i * WARM, obs_buffers=rb, zero_stats=False)
torch.cuda.synchronize()
L = nat.load()
out = (C.c_ulonglong * 32)()
L.bg_debug_counters.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
L.bg_debug_counters(env._h, out)
env.set_profiling(True)
env.rollout(T, policy=POLICY_CYCLE3, policy_seed=POLICY_SEED, t0=3 * WARM, obs_buffers=rb, zero_stats=False)
torch.cuda.synchronize()
p = env.get_profile()
L.bg_debug_counters(env._h, out)
o = [float(v) for v in out]
epw = int(os.environ.get("BG_E3_EPW", "0")) or 256
wgs = (n + epw - 1) // epw
it = max(o[4], 1)
print(f"launch {p['rollout_ms'] * 1e3:.0f} us, T {T}, {n * T / (p['rollout_ms'] * 1e-3) / 1e9:.2f} G env-steps/s")
print(f"OWNER (4 waves per workgroup): {it / (4 * wgs):.0f} iterations per wave; per iteration: step phase {o[0] / it:.0f} cycles, copy-out {o[1] / it:.0f}, idle {o[2] / it:.0f}; records per iteration {o[5] / it:.1f}")
for c, name in ((0, "play"), (1, "other")):
    b = max(o[12 + c], 1)
    print(f"SERVICE {name}: {b / wgs / T:.2f} batches per workgroup-step of {o[14 + c] / b:.1f} envs, {o[9 + c] / b:.0f} cycles each")
print(f"SERVICE waves: polling / claiming {o[8] / max(o[8] + o[9] + o[10], 1):.2f} of their time")
for k in range(3):
    c = o[16 + 4 * k + 2]
    if c:
        print(f"OWNER waves with {k + 1} engine wave(s) on their SIMD: {c / wgs:.2f} per workgroup, end {o[16 + 4 * k] / c / 100:.0f} us after the workgroup's start, {o[16 + 4 * k + 1] / c:.0f} iterations")
print(f"WORKGROUP timeline (mean, us after its first instruction): prologue done {o[28] / wgs / 100:.1f}, owner loops end {o[29] / (4 * wgs) / 100:.1f}, service loops end {o[30] / (3 * wgs) / 100:.1f}, workgroup end {o[31] / wgs / 100:.1f}")
