#!/usr/bin/env python3
"""Development: the timeline of ONE-STEP launches of bg_engine_kernel (bg_step / bg_step_rows) -- where inside a workgroup the ~16 us of a launch go.
Build with  tools/build_variant.sh engtl -DBG_ENG_TL , then  BALATRO_MI355X_LIB=build/variants/engtl.so python tools/eng_timeline.py [rows|keys]
Times are 10 ns ticks of the constant clock (wall_clock64), summed over workgroups by the kernel and divided here."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from balatro_gym_amd import BalatroVecEnv, _native as nat
from balatro_gym_amd.vec_env import ObsBuffers
from bench import jokers_for, POLICY_CYCLE3, POLICY_SEED
layout = sys.argv[1] if len(sys.argv) > 1 else "rows"
n, K, W = 65536, 200, 100
dev = torch.device("cuda", 0)


def make(**kw):
    e = BalatroVecEnv(n, [1000 + g for g in range(n)], device=0, scorer_jokers=True, autoreset=True, max_ante=4, **kw)
    e.inject(jokers=[jokers_for(g) for g in range(n)], apply_now=True)
    return e


twin = make()
acts = torch.zeros((W + K + 1, n), dtype=torch.int32, device=dev)
rec_ob = ObsBuffers(n, dev, steps=43)   # (per-step buffers: without them every step writes row 0 of `actions`)
for c0 in range(0, W + K + 1, 43):
    twin.rollout(43, policy=POLICY_CYCLE3, policy_seed=POLICY_SEED, env_index0=0, t0=c0, obs_buffers=rec_ob, actions=acts[c0:c0 + 43], zero_stats=c0 == 0)
print("twin:", twin.stats())   # (also waits for the recording rollout)
twin.close()
del rec_ob
print("recorded actions of step", W, ":", torch.bincount(acts[W].long().clamp(0, 59), minlength=60).tolist())
env = make(obs_layout=layout)
L = nat.load()
out = (C.c_ulonglong * 32)()
L.bg_debug_counters.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
for k in range(W):
    env.step(acts[k])
torch.cuda.synchronize()
L.bg_debug_counters(env._h, out)
env.set_profiling(True)
for k in range(W, W + K):
    env.step(acts[k])
torch.cuda.synchronize()
env.check()
p = env.get_profile()
L.bg_debug_counters(env._h, out)
o = [float(v) for v in out]
if os.environ.get("PROBES"):   # a -DBG_TIMING build: the BG_PROBE cycle sums of the service steps (first active lane of each batch), per workgroup and launch
    names = {1: "boss checks + hand base (since classify)", 2: "chain: Bloodstone words", 3: "chain: main-phase words ready + skip", 5: "gather selected cards", 6: "classify",
             7: "chain: individual", 8: "chain: peeks", 13: "chain: -", 14: "chain: main", 15: "hand base + joker chain (whole call)", 16: "final score + card-state effects",
             17: "boss scoring ratio", 18: "shop: stream window (loads + twist)", 10: "progress, counters, boss bookkeeping", 11: "reward shaping",
             12: "outcome (advance round / draw / boss)", 20: "play dispatch total", 21: "other dispatch total", 22: "shop inventory", 23: "service: state load + unpack",
             24: "service: cap + reset", 25: "service: mask", 26: "service: image build", 27: "service: pack + state store"}
    print(f"obs_layout {layout}: {K} one-step launches, kernel {p['step_ms'] / K * 1e3:.2f} us per launch; cycles per workgroup and launch (2.4 GHz: 2 400 = 1 us)")
    for k in sorted(names):
        print(f"  probe {k:2d} {names[k]:44s} {o[k] / (n / 256) / K:9.0f}")
    env.close()
    sys.exit(0)
wgs = max(o[0], 1.0)
us = lambda v, c=None: v / (c if c else wgs) / 100.0
print(f"obs_layout {layout}: {K} one-step launches, kernel {p['step_ms'] / K * 1e3:.2f} us per launch (its own timestamps); {wgs / K:.0f} workgroups per launch")
print(f"WORKGROUP (mean, us after its first instruction): tables loaded {us(o[1]):.2f}, phase 1 done (thread 0) {us(o[2]):.2f}, behind the barrier {us(o[3]):.2f}, "
      f"phase 2 done (thread 0) {us(o[4]):.2f}; worker wave 0 leaves its loop {us(o[9]):.2f}, the last worker wave {us(o[10]):.2f}, copier 0 {us(o[11]):.2f}; "
      f"behind the epilogue barrier {us(o[12]):.2f}, workgroup end {us(o[13]):.2f}")
for b, name in ((5, "run"), (16, "play"), (20, "other")):
    c = o[b + 1]
    if c:
        print(f"  {name:5s} batches: {c / wgs:.2f} per workgroup of {o[b + 3] / c:.1f} envs, start {us(o[b], c):.2f} us after the workgroup's, {us(o[b + 2], c):.2f} us long")
print(f"  wave 4 (owns no env thread): at the loop {us(o[24]):.2f} us, first batch claimed {us(o[25]):.2f} us after {o[26] / wgs:.1f} empty polls; that batch: run {o[27] / wgs:.2f} / play {o[28] / wgs:.2f} / other {o[29] / wgs:.2f}")
print(f"  wave 4's first look at the queues: done {us(o[30]):.2f} us; saw (mean) play {(o[31] / wgs) % 1000:.1f}, run {(o[31] / wgs) // 1000 % 1000:.0f}, other {(o[31] / wgs) // 1000000:.0f}")
# one launch alone: first workgroup start -> last workgroup end against the launch's own timestamps
env.step(acts[W + K])
torch.cuda.synchronize()
p = env.get_profile()
L.bg_debug_counters(env._h, out)
span = (int(out[15]) - ((~int(out[14])) & ((1 << 64) - 1))) / 100.0
print(f"ONE launch: first workgroup's first instruction -> last workgroup's last {span:.2f} us")
env.close()
