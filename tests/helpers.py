"""Shared helpers for the parity tests: golden-trace loading and replay against any env implementation."""
from __future__ import annotations

import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")

OBS_KEYS = [
    "hand", "hand_size", "deck_size", "selected_cards", "chips_scored", "round_chips_scored", "progress_ratio",
    "mult", "chips_needed", "money", "ante", "round", "hands_left", "discards_left", "joker_count", "joker_ids",
    "joker_slots", "consumable_count", "consumables", "consumable_slots", "shop_items", "shop_costs",
    "shop_rerolls", "hand_levels", "phase", "action_mask", "hands_played", "best_hand_this_ante",
    "boss_blind_active", "boss_blind_type", "face_down_cards",
]

TRACES = ["c1_small_only", "c2_cycle3", "c3_jokers", "c5_uniform", "c5_uniform_rich", "cards_levels", "seeds_special", "consumables",
          "consumables_scorer", "boss_forced", "boss_forced_scorer"]
POLICY_SCRIPTED = 255   # the trace's actions come from its generator (oracle/gen_golden.py boss_policy), not from a counter-hash policy
BG_INFO_BEAT_BLIND, BG_INFO_FAILED = 1, 2


_trace_cache = {}


def load_trace(name):
    """All arrays of a golden trace, decompressed once (NpzFile re-inflates on every [] access)."""
    if name not in _trace_cache:
        with np.load(os.path.join(GOLD, f"trace_{name}.npz")) as z:
            _trace_cache[name] = {k: z[k] for k in z.files}
    return _trace_cache[name]


def trace_injection(tr, si):
    """Per-seed harness injection recorded in a trace -> dict."""
    nj = int(tr["inj_njokers"][si])
    inj = {
        "jokers": [int(x) for x in tr["inj_jokers"][si, :nj]],
        "money": None if tr["inj_money"][si] < 0 else int(tr["inj_money"][si]),
        "ante": None if tr["inj_ante"][si] < 0 else int(tr["inj_ante"][si]),
        "cards": [(d, int(e), int(ed), int(s)) for d, (e, ed, s) in enumerate(tr["inj_cards"][si]) if e or ed or s],
        "levels": [(ht, int(l)) for ht, l in enumerate(tr["inj_levels"][si]) if l],
        "consumables": [int(x) for x in tr["inj_cons"][si, :int(tr["inj_ncons"][si])]] if "inj_cons" in tr else [],
    }
    return inj


def assert_obs_equal(got, want, ctx):
    for k in OBS_KEYS:
        g, w = np.asarray(got[k]), np.asarray(want[k])
        assert g.shape == w.shape, f"{ctx}: obs[{k}] shape {g.shape} vs {w.shape}"
        if not np.array_equal(g, w):
            raise AssertionError(f"{ctx}: obs[{k}] got {g} want {w}")


# ---------------------------------------------------------------------- forced hands (rare hand types on purpose)
def _code(rank, suit):
    return (rank - 2) * 4 + suit


def forced_hand_cards(ht, rr):
    """Card codes (1..5 of them) that BalatroGame._classify_hand (balatro_game.py:40-93) calls hand type `ht`."""
    suits = list(range(4))
    rr.shuffle(suits)
    if ht == 8:  # straight flush (incl. the wheel A-2-3-4-5)
        lo = rr.randint(1, 10)
        ranks = [14, 2, 3, 4, 5] if lo == 1 else list(range(lo, lo + 5))
        return [_code(r, suits[0]) for r in ranks]
    if ht == 7:  # four of a kind, with or without a kicker
        r = rr.randint(2, 14)
        cards = [_code(r, s) for s in range(4)]
        if rr.random() < 0.5:
            cards.append(_code(rr.choice([x for x in range(2, 15) if x != r]), rr.randrange(4)))
        return cards
    if ht == 6:  # full house
        a, b = rr.sample(range(2, 15), 2)
        return [_code(a, s) for s in rr.sample(range(4), 3)] + [_code(b, s) for s in rr.sample(range(4), 2)]
    if ht == 5:  # flush that is not a straight
        while True:
            ranks = sorted(rr.sample(range(2, 15), 5))
            if ranks[4] - ranks[0] != 4 and ranks != [2, 3, 4, 5, 14]:
                return [_code(r, suits[0]) for r in ranks]
    if ht == 4:  # straight in at least two suits
        lo = rr.randint(1, 10)
        ranks = [14, 2, 3, 4, 5] if lo == 1 else list(range(lo, lo + 5))
        ss = [suits[0], suits[1]] + [rr.randrange(4) for _ in range(3)]
        return [_code(r, s) for r, s in zip(ranks, ss)]
    if ht == 3:
        r = rr.randint(2, 14)
        return [_code(r, s) for s in rr.sample(range(4), 3)]
    if ht == 2:
        a, b = rr.sample(range(2, 15), 2)
        return [_code(a, s) for s in rr.sample(range(4), 2)] + [_code(b, s) for s in rr.sample(range(4), 2)]
    if ht == 1:
        r = rr.randint(2, 14)
        return [_code(r, s) for s in rr.sample(range(4), 2)]
    return [_code(rr.randint(2, 14), rr.randrange(4))]


def forced_deck(ht, rr):
    """A 52-card deck whose first k cards form hand type `ht` (the hand after the blind choice is deck[0..7], and the
    classifier reads deck[position]): returns (deck codes, k)."""
    head = forced_hand_cards(ht, rr)
    rest = [c for c in range(52) if c not in head]
    rr.shuffle(rest)
    return head + rest, len(head)


def forced_hand_script(k, rr, blind=45):
    """Actions of a forced-hand episode start: choose the blind, select positions 0..k-1 (in a random order), play."""
    order = list(range(k))
    rr.shuffle(order)
    return [blind] + [2 + p for p in order] + [0]


# ---------------------------------------------------------------------------------------------------------
# Workloads of the sharded (N > 1) tests, by GLOBAL env index: a shard injects its own slice and must reproduce the bytes
# the one-process run holds for those envs.
#   "configs2": BASELINE configs[2] -- scorer-level joker chain, Antes 1-4 cap (the bench workload)
#   "configs3": BASELINE configs[3] style -- card states (enhancement / edition / seal on half the deck), jokers by id and
#               two consumables per episode out of all tarots / planets / spectrals
# ---------------------------------------------------------------------------------------------------------
CONSUMABLE_POOL = list(range(1, 23)) + list(range(30, 42)) + list(range(50, 68))


def sharded_workload(kind, total, seed0=6000):
    import random
    seeds = [seed0 + i for i in range(total)]
    if kind == "configs2":
        return dict(seeds=seeds, env_kwargs=dict(autoreset=True, scorer_jokers=True, max_ante=4), jokers=None, cards=None, consumables=None)
    assert kind == "configs3", kind
    jokers = [random.Random(seed0 + i).sample(range(1, 151), i % 6) for i in range(total)]
    cards = []
    for i in range(total):
        rr = random.Random(seed0 * 7 + i)
        cards.append([(d, rr.choice([0, 1, 4, 5, 6, 7, 8]), rr.choice([0, 0, 1, 2]), rr.choice([0, 0, 1, 2, 3, 4])) for d in rr.sample(range(52), 26)])
    cons = [random.Random(seed0 * 13 + i).sample(CONSUMABLE_POOL, 1 + (i % 5 != 0)) for i in range(total)]
    return dict(seeds=seeds, env_kwargs=dict(autoreset=True, scorer_jokers=True, max_ante=4, card_states=True), jokers=jokers, cards=cards, consumables=cons)


def apply_sharded_workload(env, wl, lo, hi):
    """env: the BalatroVecEnv of envs [lo, hi) of the workload."""
    if wl["jokers"] is not None:
        env.inject(jokers=wl["jokers"][lo:hi], apply_now=True)
    if wl["cards"] is not None:
        env.inject_cards(wl["cards"][lo:hi], apply_now=True)
    if wl["consumables"] is not None:
        env.inject_consumables(wl["consumables"][lo:hi], apply_now=True)
