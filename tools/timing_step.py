#!/usr/bin/env python3
"""Development aid: cycle counters of the step engine behind bg_step (one launch per step, caller's actions, per-key output, info arrays).
Build with tools/build_variant.sh t4 -DBG_TIMING4 and run with BALATRO_MI355X_LIB=build/variants/t4.so."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from balatro_gym_amd import BalatroVecEnv, _native as nat
from bench import jokers_for, POLICY_CYCLE3, POLICY_SEED
n, K = 65536, int(os.environ.get("K", "100"))
def make():
    e = BalatroVecEnv(n, [1000 + g for g in range(n)], device=0, scorer_jokers=True, autoreset=True, max_ante=4)
    e.inject(jokers=[jokers_for(g) for g in range(n)], apply_now=True)
    return e
twin = make()
acts = torch.zeros((K, n), dtype=torch.int32, device=twin.device)
twin.rollout(400, policy=POLICY_CYCLE3, policy_seed=POLICY_SEED)            # mixed phases first
env = make(); env.rollout(400, policy=POLICY_CYCLE3, policy_seed=POLICY_SEED)
twin.rollout(K, policy=POLICY_CYCLE3, policy_seed=POLICY_SEED, t0=400, actions=acts)
twin.close()
L = nat.load()
out = (C.c_ulonglong * 32)()
L.bg_debug_counters.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
for k in range(10): env.step(acts[k])
torch.cuda.synchronize(); L.bg_debug_counters(env._h, out)
env.set_profiling(True)
for k in range(10, K): env.step(acts[k])
torch.cuda.synchronize()
p = env.get_profile(); L.bg_debug_counters(env._h, out); o = list(out)
nl = K - 10; wg = n / 256
print(f"bg_step: {p['step_ms']/nl*1e3:.1f} us per launch (kernel); per launch and workgroup: prologue {o[16]/wg/nl/2400:.2f} us, worker waves {o[1]/wg/nl:.1f}, wave cycles {o[0]/max(1,o[1]):.0f} ({o[0]/max(1,o[1])/2400:.1f} us), idle {o[11]/max(1,o[0]):.2f}")
for c, nm in ((0, "run"), (1, "play"), (2, "other")):
    b = max(1, o[2 + 3 * c])
    print(f"  {nm:5s}: batches per workgroup-launch {o[2+3*c]/wg/nl:.2f}  items per batch {o[3+3*c]/b:.1f}  cycles per batch {o[4+3*c]/b:.0f} ({o[4+3*c]/b/2400:.1f} us)")
env.close()
