"""Development aid: first divergence between the fused rollout and the oracle in the consumables test configuration."""
import random
import sys

import numpy as np

sys.path.insert(0, ".")
from tests.test_gpu_parity import _vec, _oracle_rollout  # noqa: E402
from tests.helpers import OBS_KEYS  # noqa: E402
from balatro_gym_amd.vec_env import RowBuffers  # noqa: E402

scorer = bool(int(sys.argv[1])) if len(sys.argv) > 1 else True
n, T = 256, 160
seeds = [93_000 + 5 * i for i in range(n)]
pool = list(range(1, 23)) + list(range(30, 42)) + list(range(50, 68))
jokers = [random.Random(5000 + i).sample(list(range(1, 151)), i % 6) for i in range(n)]
cons = [[pool[i % len(pool)], random.Random(6000 + i).choice(pool)][: 2 - (i % 9 == 0)] for i in range(n)]
cards = []
for i in range(n):
    rr = random.Random(7000 + i)
    cards.append([(d, rr.choice([0, 0, 4, 8]), 0, rr.choice([0, 3, 4, 4])) for d in rr.sample(range(52), 16)] if i % 2 else [])
env = _vec(n, seeds, scorer_jokers=scorer, autoreset=True, max_ante=6, card_states=True)
env.inject(jokers=jokers, apply_now=True)
env.inject_cards(cards, apply_now=True)
env.inject_consumables(cons, apply_now=True)
rb = RowBuffers(n, env.device, steps=T)
env.rollout(T, policy=0, policy_seed=47, obs_buffers=rb)
env.check()
wobs, wr, wt, wa, wstats = _oracle_rollout(n, seeds, T, 0, 47, scorer, 6, jokers, cards=cards, consumables=cons)
ga = rb.action.cpu().numpy(); gr = rb.reward.contiguous().cpu().numpy(); gt = rb.terminated.cpu().numpy()
gobs = {k: rb.tensors[k].contiguous().cpu().numpy() for k in OBS_KEYS}
bad = np.zeros((T, n), bool)
bad |= ga != wa
bad |= gr.view(np.uint64) != wr.view(np.uint64)
bad |= gt != wt
for k in OBS_KEYS:
    bad |= (gobs[k] != wobs[k]).reshape(T, n, -1).any(axis=2)
envs = np.nonzero(bad.any(axis=0))[0]
print("diverging envs:", envs.tolist())
for i in envs[:6]:
    t = int(np.argmax(bad[:, i]))
    print(f"--- env {i} first divergence at t={t}  injected cons={cons[i]} jokers={jokers[i]}")
    for tt in range(max(0, t - 3), t + 1):
        print(f"  t={tt} action got {ga[tt, i]} want {wa[tt, i]} reward got {gr[tt, i]} want {wr[tt, i]} term {gt[tt, i]}/{wt[tt, i]}")
        for k in OBS_KEYS:
            if not np.array_equal(gobs[k][tt, i], wobs[k][tt, i]):
                print(f"     obs[{k}] got {gobs[k][tt, i].tolist()} want {wobs[k][tt, i].tolist()}")
        print(f"     consumables(want) {wobs['consumables'][tt, i].tolist()} jokers {wobs['joker_ids'][tt, i].tolist()} money {wobs['money'][tt, i]} phase {wobs['phase'][tt, i]}")
