#!/usr/bin/env python3
"""Development aid: cycle counters of the service-wave rollout kernel (build with -DBG_TIMING3, BG_ROLLOUT_V=3)."""
import ctypes as C, os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from balatro_gym_amd import BalatroVecEnv, _native as nat
from balatro_gym_amd.vec_env import RowBuffers
from bench import jokers_for, POLICY_CYCLE3, POLICY_SEED
n, T = 65536, int(os.environ.get("T", "372"))
env = BalatroVecEnv(n, [1000 + g for g in range(n)], device=0, scorer_jokers=True, autoreset=True, max_ante=4)
env.inject(jokers=[jokers_for(g) for g in range(n)], apply_now=True)
rb = RowBuffers(n, env.device, steps=T)
for i in range(3):
    env.rollout(T, policy=POLICY_CYCLE3, policy_seed=POLICY_SEED, t0=i * T, obs_buffers=rb, zero_stats=False)
torch.cuda.synchronize()
L = nat.load()
out = (C.c_ulonglong * 32)()
L.bg_debug_counters.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
L.bg_debug_counters(env._h, out)
env.set_profiling(True)
env.rollout(T, policy=POLICY_CYCLE3, policy_seed=POLICY_SEED, t0=3 * T, obs_buffers=rb, zero_stats=False)
torch.cuda.synchronize()
p = env.get_profile()
L.bg_debug_counters(env._h, out)
o = list(out)
waves = max(1, o[11]); iters = max(1, o[1])
print(f"launch {p['rollout_ms']*1e3:.0f} us, env waves {waves}, T {T}")
print(f"env wave: cycles {o[0]/waves:.0f}  iterations {o[1]/waves:.1f} ({o[1]/waves/T:.2f} per step)  idle iterations {o[2]/waves:.1f}"
      f"  lanes finishing per iteration {o[12]/iters:.1f}")
print(f"  cycles per iteration: A {o[3]/iters:.0f}  C {o[4]/iters:.0f}  total {o[0]/iters:.0f}")
print(f"  phase C per iteration: poll+merge {o[16]/iters:.0f}  cap+reset {o[17]/iters:.0f}  mask {o[18]/iters:.0f}  record {o[19]/iters:.0f}"
      f"  stats+loop {(o[4]-o[16]-o[17]-o[18]-o[19])/iters:.0f}")
print(f"  phase A per iteration: policy {o[20]/iters:.0f}  guards+cheap actions {o[21]/iters:.0f}  pack+enqueue+rest {(o[3]-o[20]-o[21])/iters:.0f}")
for cls, nm in ((0, "plays"), (1, "others")):
    b = max(1, o[5 + 3 * cls])
    print(f"service {nm}: batches per wave {o[5+3*cls]/(waves/2):.1f}  items per batch {o[6+3*cls]/b:.1f}  cycles per batch {o[7+3*cls]/b:.0f}"
          f"  busy {o[7+3*cls]/max(1,o[13+cls]):.2f} of its time")
env.close()
