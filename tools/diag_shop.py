#!/usr/bin/env python3
"""Development: tests/test_gpu_operators.py::test_forced_rare_hands_vs_oracle[False] with the mismatching env's history and ring slot printed."""
import os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from balatro_gym_amd import BalatroVecEnv
from oracle import pyoracle as po
from tests.helpers import forced_deck, forced_hand_script, OBS_KEYS
scorer = False
n = 9 * 64
seeds = [123_000 + 11 * i for i in range(n)]
rr = random.Random(77)
hts = [i % 9 for i in range(n)]
decks, scripts = [], []
for i in range(n):
    d, k = forced_deck(hts[i], rr)
    decks.append(d)
    scripts.append(forced_hand_script(k, rr, blind=[45, 46, 47][(i // 9) % 3]))
jokers = [[113, 40, 33][: i % 4] for i in range(n)]
antes = [1 + (i // 27) % 8 for i in range(n)]
env = BalatroVecEnv(n, seeds, device=0, scorer_jokers=scorer, autoreset=False, max_ante=0)
env.inject(jokers=jokers, ante=antes, apply_now=True)
env.inject_deck(decks)
orc = [po.OracleEnv(s, scorer_jokers=scorer) for s in seeds]
for o, js, a, d in zip(orc, jokers, antes, decks):
    o.set_jokers(js); o.set_ante(a); o.set_deck(d)
hist = [[] for _ in range(n)]
bad = 0
for t in range(12):
    acts = np.array([scripts[i][t] if t < len(scripts[i]) else o.policy_action(0, 31, i, t) for i, o in enumerate(orc)], dtype=np.int32)
    res = [o.step(int(a)) for o, a in zip(orc, acts)]
    env.step(torch.from_numpy(acts).to(env.device))
    got = {k: v.cpu().numpy() for k, v in env.obs.items()}
    for i in range(n):
        hist[i].append(int(acts[i]))
        w = res[i][0]
        diff = [k for k in OBS_KEYS if not np.array_equal(got[k][i], w[k])]
        if diff:
            bad += 1
            if bad <= 5:
                st = BalatroVecEnv.parse_state_blob(env.get_state(i))
                cur = st["shop_slot_current"]
                slot = st["shop_slots"][cur]
                r = random.Random(int(slot[62]))
                words = [r.getrandbits(32) for _ in range(56)]
                pk = [0] * 6
                for k in range(24):
                    pk[k >> 2] |= (words[k] >> 24) << (8 * (k & 3))
                print(f"t {t} env {i} ante {antes[i]} jokers {jokers[i]} actions {hist[i]} differing keys {diff}")
                for k in diff[:4]:
                    print(f"     {k}: got {got[k][i]} want {w[k]}")
                print("   slot", cur, "seed", int(slot[62]), "words ok", list(slot[:56]) == words, "packed ok", [int(x) for x in slot[56:62]] == pk, [hex(int(x)) for x in slot[56:64]], [hex(x) for x in pk])
                for dlt in (-1, 1, 2):
                    sl = st["shop_slots"][(cur + dlt) % st["KS"]]
                    print("      slot", (cur + dlt) % st["KS"], "tail", [hex(int(x)) for x in sl[56:64]])
    if bad:
        break
    wt = np.array([r[2] for r in res], dtype=np.uint8)
    if wt.any():
        print("t", t, "resets:", np.nonzero(wt)[0].tolist()[:20])
        for i in np.nonzero(wt)[0]:
            orc[i].reset(); orc[i].set_jokers(jokers[i]); orc[i].set_ante(antes[i])
        env.reset(mask=torch.from_numpy(wt).to(env.device))
print("differing envs:", bad)
try:
    env.check()
except Exception as ex:
    print(ex)
