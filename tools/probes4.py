#!/usr/bin/env python3
"""Development aid: BG_PROBE cycle probes of the play path inside the step engine (build: tools/build_variant.sh pr -DBG_TIMING)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from balatro_gym_amd import BalatroVecEnv, _native as nat
from balatro_gym_amd.vec_env import RowBuffers
from bench import jokers_for, POLICY_CYCLE3, POLICY_SEED
n, T = int(os.environ.get("N", "65536")), int(os.environ.get("T", "372"))
env = BalatroVecEnv(n, [1000 + g for g in range(n)], device=0, scorer_jokers=True, autoreset=True, max_ante=4)
env.inject(jokers=[jokers_for(g) for g in range(n)], apply_now=True)
WARM = int(os.environ.get("WARM", str(T)))
rb = RowBuffers(n, env.device, steps=max(T, WARM), row_stride=384)
for i in range(2):
    env.rollout(WARM, policy=POLICY_CYCLE3, policy_seed=POLICY_SEED, t0=i * WARM, obs_buffers=rb, zero_stats=False)
torch.cuda.synchronize()
L = nat.load()
out = (C.c_ulonglong * 32)()
L.bg_debug_counters.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
L.bg_debug_counters(env._h, out)
env.set_profiling(True)
env.rollout(T, policy=POLICY_CYCLE3, policy_seed=POLICY_SEED, t0=2 * WARM, obs_buffers=rb, zero_stats=True)
torch.cuda.synchronize()
prof = env.get_profile()
st = env.stats()
L.bg_debug_counters(env._h, out)
o = list(out)
wgs = n / 256
names = {1: "boss checks + hand base (since classify)", 2: "chain: Bloodstone words (since chain: individual)", 3: "chain: main-phase words ready + skip", 5: "gather selected cards", 6: "classify", 7: "chain: individual", 8: "chain: peeks", 13: "chain: -", 14: "chain: main", 15: "hand base + joker chain (whole call)", 16: "final score + card-state effects", 17: "boss scoring ratio", 18: "shop: stream window (loads + twist)", 10: "progress, counters, boss bookkeeping",
         11: "reward shaping", 12: "outcome (advance round / draw / boss)", 20: "play dispatch total", 21: "other dispatch total", 22: "shop inventory",
         28: "reset: cold stores", 29: "reset: ring deck copy", 30: "reset: template loads + apply", 23: "service: state load + unpack", 24: "service: cap + reset", 25: "service: mask", 26: "service: image build", 27: "service: pack + state store"}
plays = max(1, st["plays"])
print(f"N {n} T {T} BG_E3_EPW {os.environ.get('BG_E3_EPW', '-')}: launch {prof['rollout_ms'] * 1e3:.0f} us, {st['plays']} accepted plays, {st['episodes']} episodes; cycles per workgroup-step | per accepted play (first active lane of each batch)")
for k in sorted(names):
    print(f"  probe {k:2d} {names[k]:40s} {o[k]/wgs/T:9.0f} | {o[k]/plays:9.0f}")
env.close()
