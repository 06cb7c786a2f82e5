#!/usr/bin/env python3
"""Development aid: a longer randomized parity run than the test suite (fused rollout, packed records, vs the C oracle)."""
import os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from balatro_gym_amd import BalatroVecEnv
from balatro_gym_amd.vec_env import RowBuffers
from tests.helpers import OBS_KEYS
from tests.test_gpu_parity import _oracle_rollout
from bench import IMPLEMENTED

POOL = list(range(1, 23)) + list(range(30, 42)) + list(range(50, 68))

def run(n, T, policy, scorer, cards_on, seed0, max_ante, cons_on=False):
    seed0 += int(os.environ.get("SEED_OFFSET", "0"))  # other games than the fixed set
    seeds = [seed0 + 11 * i for i in range(n)]
    jokers = [random.Random(seed0 + i).sample(IMPLEMENTED if i % 2 else list(range(1, 151)), i % 6) for i in range(n)] if scorer else None
    cards = None
    if cards_on:
        cards = []
        for i in range(n):
            rr = random.Random(seed0 * 7 + i)
            cards.append([(d, rr.choice([0, 1, 4, 5, 6, 7, 8]), rr.choice([0, 0, 1, 2]), rr.choice([0, 0, 1, 2, 3, 4] if cons_on else [0, 0, 1, 2, 3])) for d in rr.sample(range(52), 26)])
    env = BalatroVecEnv(n, seeds, scorer_jokers=scorer, autoreset=True, max_ante=max_ante, card_states=cards_on)
    if jokers:
        env.inject(jokers=jokers, apply_now=True)
    if cards:
        env.inject_cards(cards, apply_now=True)
    cons = None
    if cons_on:  # two consumables per episode out of all 52 ids (tarots, planets, spectrals)
        cons = [random.Random(seed0 * 13 + i).sample(POOL, 1 + (i % 5 != 0)) for i in range(n)]
        env.inject_consumables(cons, apply_now=True)
    rb = RowBuffers(n, env.device, steps=T, row_stride=int(os.environ.get("STRIDE", "0")))   # STRIDE=384: the whole-line record layout
    t = time.time()
    chunks = [int(c) for c in os.environ.get("CHUNKS", "").split(",") if c]   # CHUNKS=20,13,30: the T steps as that cycle of SHORT launches (the refill then goes in pieces)
    if chunks:
        done, k = 0, 0
        while done < T:
            c = min(chunks[k % len(chunks)], T - done)
            part = RowBuffers.__new__(RowBuffers)
            part.n, part.steps, part.rows = n, c, rb.rows[done:done + c]
            if hasattr(rb, "row_stride"):
                part.row_stride = rb.row_stride
            env.rollout(c, policy=policy, policy_seed=seed0, t0=done, obs_buffers=part, zero_stats=(done == 0))
            done += c; k += 1
    else:
        env.rollout(T, policy=policy, policy_seed=seed0, obs_buffers=rb)
    env.check()
    st = env.stats()
    wobs, wr, wt, wa, wst = _oracle_rollout(n, seeds, T, policy, seed0, scorer, max_ante, jokers, cards=cards, consumables=cons)
    assert np.array_equal(rb.action.cpu().numpy(), wa)
    assert np.array_equal(rb.terminated.cpu().numpy(), wt)
    assert np.array_equal(rb.reward.contiguous().cpu().numpy().view(np.uint64), wr.view(np.uint64))
    for k in OBS_KEYS:
        assert np.array_equal(rb.tensors[k].contiguous().cpu().numpy(), wobs[k]), k
    for k in ("steps", "episodes", "plays", "score_sum", "reward_bits"):
        assert st[k] == wst[k], k
    env.close()
    print(f"ok n={n} T={T} policy={policy} scorer={scorer} cards={cards_on} episodes={st['episodes']} plays={st['plays']} ({time.time()-t:.0f} s)", flush=True)

if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "big":
        run(8192, 384, 0, True, True, 987001, 8)
        run(8192, 384, 2, True, False, 987002, 4)
        print("BIG STRESS OK")
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "long":  # several launches: the refill of one runs beside the next, rings wrap
        run(2048, 1500, 2, True, False, 987005, 4)
        run(1024, 1200, 0, True, True, 987006, 8)
        print("LONG STRESS OK")
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "short":  # the same games as MANY short launches: three refill periods in pieces, rings wrap
        os.environ["CHUNKS"] = "20,13,30,7,20,20"
        run(4096, 1300, 2, True, False, 987007, 4)
        run(2048, 1100, 0, True, True, 987008, 8, cons_on=True)
        os.environ["CHUNKS"] = "20"
        run(8192, 800, 2, True, False, 987009, 4)
        print("SHORT STRESS OK")
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "wide":  # configurations drawn from a seed (argv[2]): flags, policy, caps, sizes
        rr = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
        for k in range(int(sys.argv[3]) if len(sys.argv) > 3 else 6):
            cards_on = rr.random() < 0.5
            run(rr.randint(200, 1500), rr.randint(100, 450), rr.choice([0, 0, 1, 2]), rr.random() < 0.7, cards_on,
                rr.randrange(10 ** 6, 10 ** 9), rr.choice([0, 2, 4, 6, 8]), cons_on=cards_on and rr.random() < 0.6)
        print("WIDE STRESS OK")
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "consumables":
        run(4096, 400, 0, True, True, 987003, 8, cons_on=True)
        run(4096, 400, 0, False, True, 987004, 0, cons_on=True)
        print("CONSUMABLES STRESS OK")
        sys.exit(0)
    run(1000, 300, 0, True, False, 123457, 6)
    run(1000, 300, 2, True, False, 223457, 4)
    run(777, 260, 0, True, True, 323457, 8)
    run(1500, 200, 0, False, False, 423457, 0)
    run(1000, 300, 1, False, False, 523457, 0)
    print("STRESS OK")
