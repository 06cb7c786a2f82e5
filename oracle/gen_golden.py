#!/usr/bin/env python3
"""Generate the committed golden vectors under tests/golden/ by importing the Python reference.

BUILD-CONTAINER ONLY (needs /root/reference); TEST INFRASTRUCTURE.  Run:  python3 oracle/gen_golden.py
Only DATA (inputs + the reference's outputs) is written; no reference source travels.

Files:
  F1 mt_streams.json   CPython random.Random known answers + shuffled decks from DeterministicRNG stream 0
  F2 classify.npz      random 1..8-card subsets -> BalatroGame._classify_hand; checksum over all C(52,5) hands
  F3 score_hand.json   UnifiedScorer.score_hand cases with joker NAME lists (operator-level joker chain)
  F3b score_hand_dups.json  the same with REPEATED jokers (Ankh's copies): several Bloodstones / 8 Balls drawing per card
  F4 trace_<cfg>.npz   BalatroEnv traces under the counter-hash policy: actions, rewards, terminated, info and the
                       full observation after every step
  F5 trace_boss_forced[_scorer].npz  every BossBlindType forced as the first boss blind of 24 seeds (the seed picks it), boss-only policy
  F6 sim_eval.npz      balatro_sim.BalatroSimulator: evaluate_hand (10 000 hands, with / without Four Fingers / Shortcut) and
                       calculate_score (2 000 cases: enhancements, editions, seals, joker-major chain, global-stream position)
  F8 sb3_fixed.npz     SafeBalatroEnv(BalatroEnvFixed(seed)) (train_balatro_fixed.py) stepped like a VecEnv: fixed space + observations
  F7 kat.json          the reference's own known answers (tests/chips_test.py:5-24, balatro_trajectories.json)
"""
from __future__ import annotations

import json
import os
import random
import sys
import zlib
from itertools import combinations

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import refharness as rh  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")

OBS_KEYS = [
    "hand", "hand_size", "deck_size", "selected_cards", "chips_scored", "round_chips_scored", "progress_ratio",
    "mult", "chips_needed", "money", "ante", "round", "hands_left", "discards_left", "joker_count", "joker_ids",
    "joker_slots", "consumable_count", "consumables", "consumable_slots", "shop_items", "shop_costs",
    "shop_rerolls", "hand_levels", "phase", "action_mask", "hands_played", "best_hand_this_ante",
    "boss_blind_active", "boss_blind_type", "face_down_cards",
]

# the 51 jokers complete_joker_effects.py implements (ids from jokers.py)
IMPLEMENTED = [1, 136, 27, 38, 61, 16, 34, 108, 23, 22, 53, 97, 50, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15,
               131, 132, 133, 134, 135, 48, 128, 122, 72, 140, 31, 39, 40, 41, 101, 124, 26, 33, 104, 147, 118, 119,
               116, 117]


def gen_mt():
    out = {"seeds": [], "u32": [], "randbelow": [], "random": [], "deck": [], "shop_seed": []}
    ref = rh.load_reference()
    seeds = [1, 7, 42, 382, 2 ** 32 - 1, 2 ** 32 + 5, 10 ** 12, 123456789, 0x7FFFFFFF, 2 ** 31, 1000, 1001, 65535,
             2 ** 40 + 3, 999999937, 31337] + [random.Random(99).randrange(1, 2 ** 48) for _ in range(48)]
    for s in seeds:
        r = random.Random(s)
        out["seeds"].append(s)
        out["u32"].append([r.getrandbits(32) for _ in range(16)])
        r = random.Random(s)
        out["randbelow"].append([[n, r._randbelow(n)] for n in (52, 3, 145, 2, 28, 24, 2 ** 31, 51, 8, 7, 6, 2 ** 32 - 1)])
        r = random.Random(s)
        out["random"].append([r.random().hex() for _ in range(8)])
        rng = ref["env2"].DeterministicRNG(s)
        deck = [ref["cards"].Card(rank=rk, suit=su) for su in ref["cards"].Suit for rk in ref["cards"].Rank]
        rng.shuffle("deck_shuffle", deck)
        out["deck"].append([int(c) for c in deck])
        out["shop_seed"].append([rng.get_int("shop_generation", 0, 2 ** 31 - 1) for _ in range(3)])
    with open(os.path.join(GOLD, "mt_streams.json"), "w") as f:
        json.dump(out, f)
    print("F1 mt_streams.json", len(seeds), "seeds")


def gen_classify():
    ref = rh.load_reference()
    Card, Rank, Suit = ref["cards"].Card, ref["cards"].Rank, ref["cards"].Suit
    game = ref["bg"].BalatroGame()
    allc = [Card(rank=Rank(c // 4 + 2), suit=Suit(c % 4)) for c in range(52)]
    r = random.Random(2024)
    N = 20000
    cards = np.full((N, 8), 255, dtype=np.uint8)
    n = np.zeros(N, dtype=np.uint8)
    ht = np.zeros(N, dtype=np.uint8)
    for i in range(N):
        k = r.randint(1, 8)
        if i % 4 == 0:  # bias towards made hands: few ranks / one suit
            pool = [c for c in range(52) if (c // 4) in r.sample(range(13), r.randint(1, 6))] if i % 8 == 0 else \
                   [c for c in range(52) if (c % 4) == r.randrange(4) or r.random() < 0.1]
            if len(pool) < k:
                pool = list(range(52))
        else:
            pool = list(range(52))
        pick = r.sample(pool, k)
        cards[i, :k] = pick
        n[i] = k
        ht[i] = int(game._classify_hand([allc[c] for c in pick])[0])
    # all C(52,5) hands in lexicographic order -> per-type counts and a CRC32 over the type bytes
    types = bytearray()
    for combo in combinations(range(52), 5):
        types.append(int(game._classify_hand([allc[c] for c in combo])[0]))
    counts = np.bincount(np.frombuffer(bytes(types), dtype=np.uint8), minlength=12)
    np.savez_compressed(os.path.join(GOLD, "classify.npz"), cards=cards, n=n, hand_type=ht,
                        all5_counts=counts, all5_crc32=np.uint32(zlib.crc32(bytes(types))))
    print("F2 classify.npz", N, "subsets; all5 counts", counts.tolist())


def gen_score_hand_dups():
    """F3b score_hand_dups.json: joker name lists WITH repeats (Ankh copies a joker: `state.jokers` then holds an id twice), dense in the
    jokers that draw per card -- several Bloodstones, several 8 Balls, Triboulet, Rough Gem -- on hands full of Hearts and 8s."""
    ref = rh.load_reference()
    us, se, cje = ref["us"], ref["se"], ref["cje"]
    names = {j.id: j.name for j in ref["jokers"].JOKER_LIBRARY}
    ENV_NAMES = ["High Card", "One Pair", "Two Pair", "Three Kind", "Straight", "Flush", "Full House", "Four Kind",
                 "Straight Flush", "Five Kind", "Flush House", "Flush Five"]
    SUITS = ["Clubs", "Diamonds", "Hearts", "Spades", "Stone"]
    pool = [117, 117, 26, 26, 147, 116, 1, 31, 3, 27, 72, 119]
    r = random.Random(177)
    cases = []
    for i in range(1000):
        ncards = r.randint(1, 8)
        cards = []
        for _ in range(ncards):
            rank, suit = r.choice([8, 8, 8, 2, 12, 13, 14, 5]), r.choice([2, 2, 2, 0, 1, 3])
            chips = 11 if rank == 14 else min(rank, 10)
            if r.random() < 0.06:
                rank, suit, chips = 0, 4, chips + 50
            cards.append([rank, suit, chips])
        nscoring = ncards if r.random() < 0.7 else r.randint(1, ncards)
        ht, level = r.randrange(9), 1
        jokers = [r.choice(pool) for _ in range(r.randint(2, 5))]
        hands_left, discards_left, deck_len, gseed = r.randint(1, 4), r.randint(0, 3), 52, r.randrange(2 ** 32)
        engine = se.ScoreEngine()
        engine.set_hand_level(se.HandType(ht), level)
        scorer = us.UnifiedScorer(engine, cje.CompleteJokerEffects())
        objs = [type("Card", (), {"rank": c[0], "suit": SUITS[c[1]], "chip_value": (lambda v=c[2]: v)}) for c in cards]
        ctx = us.ScoringContext(cards=objs, scoring_cards=objs[:nscoring], hand_type=se.HandType(ht), hand_type_name=ENV_NAMES[ht],
                                game_state={"jokers": [names[j] for j in jokers], "hands_left": hands_left,
                                            "discards_left": discards_left, "deck": [0] * deck_len, "money": 0})
        random.seed(gseed)
        score, bd = scorer.score_hand(ctx)
        probe = random.getrandbits(32)
        cases.append({"cards": cards, "nscoring": nscoring, "style": 0, "hand_type": ht, "level": level,
                      "jokers": jokers, "hands_left": hands_left, "discards_left": discards_left,
                      "deck_len": deck_len, "gseed": gseed, "score": int(score), "chips": int(bd["final_chips"]),
                      "mult": int(bd["final_mult"]), "x_mult": float(bd["final_x_mult"]).hex(),
                      "money": int(bd["money_gained"]), "probe": probe})
    with open(os.path.join(GOLD, "score_hand_dups.json"), "w") as f:
        json.dump(cases, f, separators=(",", ":"))
    print("F3b score_hand_dups.json", len(cases), "cases,", sum(len(set(c["jokers"])) < len(c["jokers"]) for c in cases), "with a repeated joker")


def gen_score_hand():
    ref = rh.load_reference()
    us, se, cje = ref["us"], ref["se"], ref["cje"]
    names = {j.id: j.name for j in ref["jokers"].JOKER_LIBRARY}
    ENV_NAMES = ["High Card", "One Pair", "Two Pair", "Three Kind", "Straight", "Flush", "Full House", "Four Kind",
                 "Straight Flush", "Five Kind", "Flush House", "Flush Five"]
    SIM_NAMES = ["High Card", "Pair", "Two Pair", "Three of a Kind", "Straight", "Flush", "Full House",
                 "Four of a Kind", "Straight Flush", "Five of a Kind", "Flush House", "Flush Five"]
    SUITS = ["Clubs", "Diamonds", "Hearts", "Spades", "Stone"]
    r = random.Random(77)
    cases = []
    for i in range(3000):
        ncards = r.randint(1, 8)
        cards = []
        for _ in range(ncards):
            rank = r.randint(2, 14)
            suit = r.randrange(4)
            chips = 11 if rank == 14 else min(rank, 10)
            if r.random() < 0.05:
                rank, suit, chips = 0, 4, chips + 50  # stone (balatro_env_2.py:304-306)
            elif r.random() < 0.1:
                chips += r.choice([30, 50, 80])
            cards.append([rank, suit, chips])
        nscoring = ncards if r.random() < 0.7 else r.randint(1, ncards)
        style = i & 1
        ht = r.randrange(12) if r.random() < 0.2 else r.randrange(9)
        level = r.randint(1, 15) if r.random() < 0.3 else 1
        nj = r.randint(0, 5)
        pool = IMPLEMENTED if r.random() < 0.8 else list(range(1, 151))
        jokers = r.sample(pool, nj)
        hands_left, discards_left = r.randint(1, 4), r.randint(0, 3)
        deck_len = 52
        gseed = r.randrange(2 ** 32)
        engine = se.ScoreEngine()
        engine.set_hand_level(se.HandType(ht), level)
        scorer = us.UnifiedScorer(engine, cje.CompleteJokerEffects())
        objs = [type("Card", (), {"rank": c[0], "suit": SUITS[c[1]], "chip_value": (lambda v=c[2]: v)}) for c in cards]
        ctx = us.ScoringContext(cards=objs, scoring_cards=objs[:nscoring], hand_type=se.HandType(ht),
                                hand_type_name=(SIM_NAMES if style else ENV_NAMES)[ht],
                                game_state={"jokers": [names[j] for j in jokers], "hands_left": hands_left,
                                            "discards_left": discards_left, "deck": [0] * deck_len, "money": 0})
        random.seed(gseed)
        score, bd = scorer.score_hand(ctx)
        probe = random.getrandbits(32)  # identifies how far the global stream advanced
        cases.append({"cards": cards, "nscoring": nscoring, "style": style, "hand_type": ht, "level": level,
                      "jokers": jokers, "hands_left": hands_left, "discards_left": discards_left,
                      "deck_len": deck_len, "gseed": gseed, "score": int(score), "chips": int(bd["final_chips"]),
                      "mult": int(bd["final_mult"]), "x_mult": float(bd["final_x_mult"]).hex(),
                      "money": int(bd["money_gained"]), "probe": probe})
    with open(os.path.join(GOLD, "score_hand.json"), "w") as f:
        json.dump(cases, f, separators=(",", ":"))
    print("F3 score_hand.json", len(cases), "cases")


def sim_random_hand(r, n=None):
    """A hand for the balatro_sim fixtures: (rank, suit, base_value, enhancement, edition, seal) tuples.  A mix of plain draws
    from a deck, hands squeezed into few ranks / one suit, and hands with duplicate cards (Five of a Kind, Flush Five, Flush
    House only exist with duplicates; equal dataclasses also exercise `card not in flush` at balatro_sim.py:277)."""
    n = n or r.choice([1, 2, 3, 4, 5, 5, 5, 5, 5, 6, 7, 8])
    kind = r.random()
    cards = []
    if kind < 0.45:
        deck = [(rk, su) for rk in range(2, 15) for su in range(4)]
        cards = r.sample(deck, n)
    elif kind < 0.65:  # few ranks
        ranks = r.sample(range(2, 15), r.randint(1, 3))
        cards = [(r.choice(ranks), r.randrange(4)) for _ in range(n)]
    elif kind < 0.8:  # one suit (mostly), runs of ranks
        su = r.randrange(4)
        lo = r.randint(2, 10)
        pool = [14, 2, 3, 4, 5, 6] if r.random() < 0.25 else list(range(lo, min(15, lo + 6)))
        cards = [(r.choice(pool), su if r.random() < 0.9 else r.randrange(4)) for _ in range(n)]
    else:  # runs of ranks in any suit
        lo = r.randint(2, 10)
        pool = [14, 2, 3, 4, 5] if r.random() < 0.25 else list(range(lo, min(15, lo + 5)))
        r.shuffle(pool)
        cards = [(pool[i % len(pool)], r.randrange(4)) for i in range(n)]
    out = []
    for rk, su in cards:
        bv = 11 if rk == 14 else min(rk, 10)
        if r.random() < 0.1:
            bv = rk if rk <= 10 else 10  # balatro_sim._id_to_card's base values (:377-382)
        en = r.choice([0, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8]) if r.random() < 0.5 else 0
        ed = r.choice([0, 0, 1, 2, 3, 4]) if r.random() < 0.3 else 0
        se = r.choice([0, 1, 2, 3, 4]) if r.random() < 0.3 else 0
        out.append((rk, su, bv, en, ed, se))
    return out


def gen_sim():
    r = random.Random(4242)
    ev = []
    for i in range(10000):
        cards = sim_random_hand(r)
        ff, sc = (i % 4) in (1, 3), (i % 4) in (2, 3)
        top, lists = rh.sim_evaluate(cards, ff, sc)
        ev.append({"cards": cards, "ff": int(ff), "sc": int(sc), "top": top,
                   "lists": [[lists[t][0], lists[t][1]] for t in range(12)]})
    sc_cases = []
    for i in range(2000):
        cards = sim_random_hand(r, n=r.choice([1, 2, 3, 4, 5, 5, 5, 5]))
        pool = IMPLEMENTED if r.random() < 0.85 else list(range(1, 151))
        jokers = r.sample(pool, r.randint(0, 5))
        for util in (18, 69):  # Four Fingers / Shortcut (jokers.py ids): part of player_state.jokers like any other joker
            if r.random() < 0.2 and util not in jokers and len(jokers) < 5:
                jokers.insert(r.randint(0, len(jokers)), util)
        gs = None if i % 2 == 0 else {"hands_left": r.randint(1, 4), "discards_left": r.randint(0, 3)}
        deck_len = r.choice([0, 0, 40, 52])
        seed = r.randrange(2 ** 32)
        score, money, probe = rh.sim_score(cards, jokers, gs, deck_len, seed)
        sc_cases.append({"cards": cards, "jokers": jokers,
                         "hands_left": 1 if gs is None else gs["hands_left"], "discards_left": 0 if gs is None else gs["discards_left"],
                         "deck_len": deck_len, "seed": seed, "score": score, "money": money, "probe": probe})
    # the reference's own saved answers (balatro_trajectories.json play_hand transitions): 14 known scores, no jokers
    kat = json.load(open(os.path.join(GOLD, "kat.json")))["trajectories"]
    for k in kat:
        cards = [(rk, su, 11 if rk == 14 else min(rk, 10), 0, 0, 0) for rk, su in k["cards"]]
        score, _, _ = rh.sim_score(cards, [], None, 0, 1)
        assert score == k["score"], (k, score)
    # arrays (compressed): cards [M, 8, 6] (rank, suit, base_value, enhancement, edition, seal), -1 padded position lists
    def card_array(cases):
        arr = np.zeros((len(cases), 8, 6), np.int16)
        n = np.zeros(len(cases), np.int8)
        for i, c in enumerate(cases):
            n[i] = len(c["cards"])
            arr[i, :n[i]] = c["cards"]
        return arr, n
    e_cards, e_n = card_array(ev)
    e_nl = np.array([[l[0] for l in e["lists"]] for e in ev], np.int8)
    e_n0 = np.array([[len(l[1]) for l in e["lists"]] for e in ev], np.int8)
    e_pos = np.full((len(ev), 12, 8), -1, np.int8)
    for i, e in enumerate(ev):
        for t, l in enumerate(e["lists"]):
            e_pos[i, t, :len(l[1])] = l[1]
    s_cards, s_n = card_array(sc_cases)
    s_jok = np.zeros((len(sc_cases), 5), np.int32)
    for i, c in enumerate(sc_cases):
        s_jok[i, :len(c["jokers"])] = c["jokers"]
    np.savez_compressed(
        os.path.join(GOLD, "sim_eval.npz"),
        e_cards=e_cards, e_n=e_n, e_ff=np.array([e["ff"] for e in ev], np.int8), e_sc=np.array([e["sc"] for e in ev], np.int8),
        e_top=np.array([e["top"] for e in ev], np.int8), e_nlists=e_nl, e_n0=e_n0, e_pos=e_pos,
        s_cards=s_cards, s_n=s_n, s_jokers=s_jok, s_njokers=np.array([len(c["jokers"]) for c in sc_cases], np.int32),
        s_hands_left=np.array([c["hands_left"] for c in sc_cases], np.int32),
        s_discards_left=np.array([c["discards_left"] for c in sc_cases], np.int32),
        s_deck_len=np.array([c["deck_len"] for c in sc_cases], np.int32), s_seed=np.array([c["seed"] for c in sc_cases], np.uint32),
        s_score=np.array([c["score"] for c in sc_cases], np.int64), s_money=np.array([c["money"] for c in sc_cases], np.int64),
        s_probe=np.array([c["probe"] for c in sc_cases], np.uint32))
    tops = np.bincount([e["top"] for e in ev], minlength=12)
    print("F6 sim_eval.npz", len(ev), "evaluate cases (top counts", tops.tolist(), "),", len(sc_cases), "score cases;",
          len(kat), "trajectory answers reproduced")


def gen_sb3_fixed():
    """F8 sb3_fixed.npz: the reference's own wrappers -- SafeBalatroEnv(BalatroEnvFixed(seed + rank)) (train_balatro_fixed.py:20-288)
    -- stepped like an SB3 VecEnv (finished envs are reset in the same step): the fixed observation space (51 keys: dtype /
    shape), every fixed observation, float32 rewards, dones, the wrapper's info flags and its terminal observations."""
    S, T, seed0, max_inv, max_steps = 24, 120, 300, 5, 40
    envs = [rh.RefFixedEnv(seed0 + r, max_inv, max_steps) for r in range(S)]
    spaces = envs[0].fixed.observation_space.spaces
    keys = list(spaces)
    obs = [e.reset() for e in envs]
    rec = {"seed0": np.int64(seed0), "max_invalid_actions": np.int32(max_inv), "max_episode_steps": np.int32(max_steps),
           "keys": np.array(keys), "dtypes": np.array([spaces[k].dtype.name for k in keys]),
           "shapes": np.array([",".join(str(x) for x in spaces[k].shape) for k in keys]),
           "actions": np.zeros((S, T), np.int32), "rewards": np.zeros((S, T), np.float32), "dones": np.zeros((S, T), np.uint8),
           "terminated": np.zeros((S, T), np.uint8), "truncated": np.zeros((S, T), np.uint8),
           "invalid_action_termination": np.zeros((S, T), np.uint8), "max_steps_reached": np.zeros((S, T), np.uint8)}
    for k in keys:
        rec["obs0_" + k] = np.stack([o[k] for o in obs])
        rec["obs_" + k] = np.zeros((S, T) + spaces[k].shape, spaces[k].dtype)
        rec["term_" + k] = np.zeros((S, T) + spaces[k].shape, spaces[k].dtype)  # terminal observation of a finished episode
    for t in range(T):
        for i, e in enumerate(envs):
            if i % 4 == 0:
                a = 59  # never valid: SafeBalatroEnv ends the episode after max_invalid_actions of them in a row
            else:
                a = rh.policy_action(obs[i]["action_mask"], int(obs[i]["phase"][0]), rh.POLICY_UNIFORM, 77, i, t)
            o, r, term, trunc, info = e.step(a)
            rec["actions"][i, t], rec["rewards"][i, t] = a, np.float32(r)
            rec["terminated"][i, t], rec["truncated"][i, t], rec["dones"][i, t] = term, trunc, term or trunc
            rec["invalid_action_termination"][i, t] = bool(info.get("invalid_action_termination"))
            rec["max_steps_reached"][i, t] = bool(info.get("max_steps_reached"))
            if term or trunc:
                for k in keys:
                    rec["term_" + k][i, t] = o[k]
                o = e.reset()
            obs[i] = o
            for k in keys:
                rec["obs_" + k][i, t] = o[k]
    np.savez_compressed(os.path.join(GOLD, "sb3_fixed.npz"), **rec)
    print("F8 sb3_fixed.npz", S, "envs x", T, "steps;", int(rec["dones"].sum()), "episode ends,",
          int(rec["invalid_action_termination"].sum()), "by invalid actions,", int(rec["max_steps_reached"].sum()), "by the step limit,",
          int((rec["terminated"] & ~rec["invalid_action_termination"]).sum()), "game overs")


REWARD_TERMS = ["progress", "milestone", "score", "hand_quality", "efficiency", "synergy", "strategy", "ante_bonus"]
BREAKDOWN_INT = ["base_chips", "base_mult", "card_chips", "joker_chips", "joker_mult", "final_chips", "final_mult", "money_gained"]
BREAKDOWN_F64 = ["joker_x_mult", "final_x_mult"]
POLICY_SCRIPTED = 255   # the actions of the trace come from the generator (boss_policy below), not from a policy the build re-derives


def error_code(info, action):
    """The reference's info dict -> the numeric code of include/balatro_mi355x.h (BG_ERR_*).  Every message the traces meet must be
    known: an unknown one stops the generator."""
    if info.get("raised"):
        return 11
    if info.get("terminated") == "max_ante_reached":
        return 9
    if info.get("terminated") == "max_score_reached":
        return 10
    msg = info.get("error")
    if msg is None:
        return 0
    if msg == "Invalid action":
        return 1
    if msg == "Must play exactly 5 cards":
        return 2
    if msg.startswith("Cannot play "):
        return 3
    if msg.startswith("Can only play "):
        return 4
    if msg.startswith("Must play at least "):
        return 5
    if msg == "Insufficient chips for reroll":
        return 6
    if msg == "Joker slots full":
        return 7
    if 10 <= action <= 14:
        return 8   # balatro_env_2.py:1166-1169 the consumable reported no success
    raise RuntimeError(f"unmapped reference error message {msg!r} (action {action})")


def boss_policy(obs, pseed, si, t):
    """F5: always take the boss blind, leave the shop at once, otherwise uniform over the valid actions."""
    phase = int(obs["phase"])
    if phase == 2:
        return 47
    if phase == 1:
        return 31
    return rh.policy_action(obs["action_mask"], phase, rh.POLICY_UNIFORM, pseed, si, t)


def trace(cfg_name, seeds, T, policy, scorer=False, jokers_fn=None, max_ante=0, money_fn=None, ante_fn=None,
          cards_fn=None, levels_fn=None, pseed=7, cons_fn=None, policy_fn=None):
    S = len(seeds)
    rec = {
        "seeds": np.array(seeds, dtype=np.int64), "policy": np.int32(policy), "policy_seed": np.uint64(pseed),
        "scorer_jokers": np.int32(scorer), "max_ante": np.int32(max_ante),
        "actions": np.zeros((S, T), np.uint8), "rewards": np.zeros((S, T), np.float64),
        "terminated": np.zeros((S, T), np.uint8), "final_score": np.zeros((S, T), np.int64),
        "hand_type": np.full((S, T), -1, np.int8), "error": np.zeros((S, T), np.uint8),
        "inj_jokers": np.zeros((S, 5), np.int32), "inj_njokers": np.zeros(S, np.int32),
        "inj_money": np.full(S, -1, np.int64), "inj_ante": np.full(S, -1, np.int32),
        "inj_cards": np.zeros((S, 52, 3), np.uint8), "inj_levels": np.zeros((S, 12), np.uint8),
        "inj_cons": np.zeros((S, 2), np.int32), "inj_ncons": np.zeros(S, np.int32),
        # the rest of the info dict (balatro_env_2.py:894-925, :1283-1288): exact error code + message, play details, boss name
        "error_code": np.zeros((S, T), np.uint8), "error_msg": np.full((S, T), "", dtype="U48"),
        "cards_played": np.zeros((S, T), np.int8), "beat_blind": np.zeros((S, T), np.uint8), "failed": np.zeros((S, T), np.uint8),
        "reward_terms": np.zeros((S, T, 8), np.float64), "breakdown_int": np.zeros((S, T, len(BREAKDOWN_INT)), np.int64),
        "breakdown_f64": np.zeros((S, T, len(BREAKDOWN_F64)), np.float64), "boss_blind": np.full((S, T), "", dtype="U16"),
    }
    if policy_fn is not None:
        rec["policy"] = np.int32(POLICY_SCRIPTED)
    obs_rec = None
    for si, seed in enumerate(seeds):
        env = rh.RefEnv(seed, scorer_jokers=scorer, max_ante=max_ante)
        js = jokers_fn(si) if jokers_fn else []
        money = money_fn(si) if money_fn else None
        ante = ante_fn(si) if ante_fn else None
        cs = cards_fn(si) if cards_fn else []
        lv = levels_fn(si) if levels_fn else []
        co = cons_fn(si) if cons_fn else []
        rec["inj_ncons"][si] = len(co)
        rec["inj_cons"][si, :len(co)] = co
        rec["inj_njokers"][si] = len(js)
        rec["inj_jokers"][si, :len(js)] = js
        if money is not None:
            rec["inj_money"][si] = money
        if ante is not None:
            rec["inj_ante"][si] = ante
        for (i, e, d, s) in cs:
            rec["inj_cards"][si, i] = (e, d, s)
        for (ht, l) in lv:
            rec["inj_levels"][si, ht] = l

        def inject():
            if js:
                env.set_jokers(js)
            if money is not None:
                env.set_money(money)
            if ante is not None:
                env.set_ante(ante)
            for (i, e, d, s) in cs:
                env.set_card_state(i, e, d, s)
            for (ht, l) in lv:
                env.set_hand_level(ht, l)
            if co:
                env.set_consumables(co)

        inject()
        obs = env.obs()
        if obs_rec is None:
            obs_rec = {k: np.zeros((S, T) + np.asarray(obs[k]).shape, np.asarray(obs[k]).dtype) for k in OBS_KEYS}
            obs0 = {k: np.zeros((S,) + np.asarray(obs[k]).shape, np.asarray(obs[k]).dtype) for k in OBS_KEYS}
        for k in OBS_KEYS:
            obs0[k][si] = obs[k]
        for t in range(T):
            a = policy_fn(obs, pseed, si, t) if policy_fn else rh.policy_action(obs["action_mask"], int(obs["phase"]), policy, pseed, si, t)
            obs, r, term, _, info = env.step(a)
            rec["actions"][si, t] = a
            rec["error_code"][si, t] = error_code(info, a)
            rec["error_msg"][si, t] = info.get("error", "") if not info.get("raised") else ""
            rec["boss_blind"][si, t] = info.get("boss_blind", "")
            if "final_score" in info:
                rec["cards_played"][si, t] = info["cards_played"]
                rec["beat_blind"][si, t] = bool(info.get("beat_blind"))
                rec["failed"][si, t] = bool(info.get("failed"))
                rec["reward_terms"][si, t] = [info["reward_breakdown"][k] for k in REWARD_TERMS]
                rec["breakdown_int"][si, t] = [int(info["score_breakdown"][k]) for k in BREAKDOWN_INT]
                rec["breakdown_f64"][si, t] = [float(info["score_breakdown"][k]) for k in BREAKDOWN_F64]
            rec["rewards"][si, t] = r
            rec["terminated"][si, t] = term
            if "final_score" in info:
                rec["final_score"][si, t] = info["final_score"]
                rec["hand_type"][si, t] = int(info["hand_type"])
            rec["error"][si, t] = 2 if info.get("raised") else (1 if "error" in info else 0)  # 2: the reference raised
            for k in OBS_KEYS:
                obs_rec[k][si, t] = obs[k]
            if term:
                env.reset()
                inject()
                obs = env.obs()
    for k in OBS_KEYS:
        rec["obs_" + k] = obs_rec[k]
        rec["obs0_" + k] = obs0[k]
    path = os.path.join(GOLD, f"trace_{cfg_name}.npz")
    np.savez_compressed(path, **rec)
    plays = int((rec["hand_type"] >= 0).sum())
    print(f"F4 trace_{cfg_name}.npz seeds={S} T={T} plays={plays} episodes={int(rec['terminated'].sum())} "
          f"errors by code={np.bincount(rec['error_code'].ravel(), minlength=12).tolist()} size={os.path.getsize(path) / 1024:.0f} KiB")
    return rec


def gen_traces():
    # C1: the reference's own smoke policy (balatro_env_2.py:1841-1849)
    trace("c1_small_only", [42, 7, 382] + [2000 + i for i in range(21)], 400, rh.POLICY_SMALL_ONLY)
    # C2: Ante-1 three blinds (45/46/47 by env index), boss blinds draw from the per-env global stream
    trace("c2_cycle3", [1000 + i for i in range(36)], 400, rh.POLICY_CYCLE3)
    # C3: 5 implemented jokers per env, scorer-level joker semantics, ante cap 4
    trace("c3_jokers", [1000 + i for i in range(36)], 400, rh.POLICY_CYCLE3, scorer=True,
          jokers_fn=lambda i: random.Random(i).sample(IMPLEMENTED, 5), max_ante=4)
    # C5-like: uniform over ALL valid actions incl. boss / shop buy / reroll / sell, full joker id range
    trace("c5_uniform", [3000 + i for i in range(30)], 600, rh.POLICY_UNIFORM)
    trace("c5_uniform_rich", [4000 + i for i in range(18)], 500, rh.POLICY_UNIFORM, scorer=True, max_ante=20,
          money_fn=lambda i: [500, 3000, 100000][i % 3], ante_fn=lambda i: [1, 2, 5, 9, 20][i % 5],
          jokers_fn=lambda i: random.Random(100 + i).sample(list(range(1, 151)), i % 6))

    def cards_fn(i):
        rr = random.Random(500 + i)
        return [(d, rr.choice([0, 0, 1, 2, 3, 4, 5, 6, 7, 8]), rr.choice([0, 0, 1, 2, 3]), rr.choice([0, 0, 1, 2, 3, 3]))
                for d in range(12)]

    trace("cards_levels", [5000 + i for i in range(18)], 400, rh.POLICY_UNIFORM, scorer=True, max_ante=20,
          cards_fn=cards_fn, levels_fn=lambda i: [(ht, random.Random(900 + i * 13 + ht).randint(1, 15)) for ht in range(9)],
          jokers_fn=lambda i: random.Random(700 + i).sample(list(range(1, 151)), i % 6),
          ante_fn=lambda i: [1, 3, 4, 6][i % 4])


def gen_trace_seeds():
    """Q12 and the corners of `DeterministicRNG(master_seed)` (balatro_env_2.py:88,105): seed 0 (falsy: the master seed is drawn from the
    env's global stream, `random.randint(0, 2**32 - 1)`), negative seeds (truthy; stream i is seeded `(seed + 1000 i) % 2**32`, Python's
    non-negative modulo), seeds at and beyond 2**32, and the same seed twice (two envs must play the same game)."""
    seeds = [0, 0, -5, -5, -1000, -16000, -(2 ** 33) - 7, 2 ** 32, 2 ** 32 + 5, 2 ** 32 - 16000, 2 ** 40 + 3, -(2 ** 62), 2 ** 62 + 12345, 16000, 1, -1]
    trace("seeds_special", seeds, 160, rh.POLICY_UNIFORM, scorer=True, max_ante=20,
          jokers_fn=lambda i: random.Random(2100 + i).sample(IMPLEMENTED, 3))


def gen_trace_consumables():
    """Tarot / spectral / planet consumables: two injected per episode, all 52 ids (Immolate and Cryptid change
    the deck length; Blue Joker among the jokers sees it); purple seals create more tarots on discards."""
    pool = list(range(1, 23)) + list(range(30, 42)) + list(range(50, 68))

    def cons_fn(i):
        rr = random.Random(1300 + i)
        return [pool[i % len(pool)], rr.choice(pool)] if i % 7 else [pool[i % len(pool)]]

    def cards_fn(i):
        rr = random.Random(1500 + i)
        return [(d, rr.choice([0, 0, 0, 4, 8]), 0, rr.choice([0, 0, 4, 4])) for d in range(16)] if i % 3 == 0 else []

    for scorer in (False, True):
        trace("consumables_scorer" if scorer else "consumables", [(7000 if scorer else 6000) + i for i in range(52)], 320,
              rh.POLICY_UNIFORM, scorer=scorer, max_ante=20, cons_fn=cons_fn, cards_fn=cards_fn,
              jokers_fn=lambda i: ([53] if i % 6 else []) + random.Random(1700 + i).sample([j for j in range(1, 151) if j != 53], max(0, i % 6 - 1)),
              money_fn=lambda i: [None, 3, 15, 200][i % 4])


def gen_boss():
    """F5: every one of the 28 BossBlindTypes FORCED as the first boss blind of its seeds.  select_boss_blind
    (boss_blinds.py:522-532) is random.choice(list(BossBlindType)) on the env's global stream, so the type of the first boss is a function
    of the env seed: the seeds are searched (harness convention G(seed) = seed + 16000), the reference runs UNPATCHED.  Policy: always the
    boss blind (47), leave the shop at once, otherwise uniform -- later boss blinds of a trace are free-running.  Two traces: the live env
    (inert jokers) and scorer-level joker names with jokers / antes / hand levels injected.  The generator asserts the coverage the
    tests rely on: >= 20 accepted plays under every type, >= 1 rejection by every restrictive type (Psychic / Eye / Mouth / Verdant)."""
    per_type = 24
    for scorer in (False, True):
        seeds, want = [], []
        cand = 20000 if scorer else 10000
        found = {b: 0 for b in range(1, 29)}
        while min(found.values()) < per_type:
            b = 1 + random.Random(rh.global_seed(cand))._randbelow(28)
            if found[b] < per_type:
                found[b] += 1
                seeds.append(cand)
                want.append(b)
            cand += 1
        order = sorted(range(len(seeds)), key=lambda i: (want[i], seeds[i]))
        seeds, want = [seeds[i] for i in order], [want[i] for i in order]
        kw = dict(scorer=True, max_ante=20, jokers_fn=lambda i: random.Random(2100 + i).sample(list(range(1, 151)), i % 6),
                  ante_fn=lambda i: [1, 1, 2, 3, 5, 8][i % 6],
                  levels_fn=lambda i: [(ht, random.Random(2300 + i * 13 + ht).randint(1, 9)) for ht in range(9)] if i % 2 else []) if scorer else {}
        rec = trace("boss_forced_scorer" if scorer else "boss_forced", seeds, 72, rh.POLICY_UNIFORM, policy_fn=boss_policy, pseed=11, **kw)
        # coverage: the boss type in force when a step was taken = boss_blind_type of the PREVIOUS observation
        prev = np.concatenate([rec["obs0_boss_blind_type"][:, None], rec["obs_boss_blind_type"][:, :-1]], axis=1).astype(int)
        prev[:, 1:][rec["terminated"][:, :-1] != 0] = 0   # a reset precedes the next step
        assert [int(b) for b in rec["obs_boss_blind_type"][:, 0]] == want, "the first boss blind of every seed is the forced one"
        plays = np.bincount(prev[rec["hand_type"] >= 0], minlength=29)
        rejected = {c: np.bincount(prev[rec["error_code"] == c], minlength=29) for c in (2, 3, 4, 5)}
        print("   accepted plays under boss types 1..28:", plays[1:].tolist())
        print("   rejections: Psychic", int(rejected[2][7]), "Eye", int(rejected[3][12]), "Mouth", int(rejected[4][13]), "Verdant", int(rejected[5][25]))
        assert plays[1:].min() >= 20, plays
        assert rejected[2][7] and rejected[3][12] and rejected[4][13] and rejected[5][25]


def gen_kat():
    """The reference's own known answers, as data."""
    kat = {"chips_test": [
        # tests/chips_test.py:5-24 -- (cards as [rank, suit], expected chips, hand type the old evaluator used)
        {"cards": [[14, 3]] * 5, "score": 3440, "hand_type": 11},
        {"cards": [[6, 1]] * 5, "score": 3040, "hand_type": 11},
        {"cards": [[r, 3] for r in (2, 3, 4, 5, 6)], "score": 960, "hand_type": 8},
        {"cards": [[r, 1] for r in (14, 13, 12, 11, 10)], "score": 1208, "hand_type": 8},
        {"cards": [[r, 0] for r in (2, 3, 4, 5, 14)], "score": 1000, "hand_type": 8},
        {"cards": [[2, 0], [3, 0], [4, 0], [5, 0], [14, 1]], "score": 220, "hand_type": 4},
        {"cards": [[14, 3]], "score": 16, "hand_type": 0},
    ], "trajectories": []}
    suit_id = {"Clubs": 0, "Diamonds": 1, "Hearts": 2, "Spades": 3}
    with open(os.path.join(rh.REFERENCE_ROOT, "balatro_gym", "balatro_trajectories.json")) as f:
        trajs = json.load(f)
    for traj in trajs:
        for tr in traj:
            if tr["action"]["type"] != "play_hand":
                continue
            hand = tr["state"]["hand_cards"]
            played = [hand[i] for i in tr["action"]["card_indices"]]
            gained = tr["next_state"]["score"] - tr["state"]["score"]
            kat["trajectories"].append({"cards": [[c[0], suit_id[c[1]]] for c in played], "score": gained})
    with open(os.path.join(GOLD, "kat.json"), "w") as f:
        json.dump(kat, f)
    print("F7 kat.json", len(kat["chips_test"]), "+", len(kat["trajectories"]), "known answers")


def main():
    os.makedirs(GOLD, exist_ok=True)
    which = sys.argv[1:] or ["mt", "classify", "score", "traces", "consumables", "boss", "kat", "sim", "sb3"]
    if "mt" in which:
        gen_mt()
    if "classify" in which:
        gen_classify()
    if "score" in which:
        gen_score_hand()
    if "score" in which or "score_dups" in which:
        gen_score_hand_dups()
    if "traces" in which:
        gen_traces()
    if "consumables" in which:
        gen_trace_consumables()
    if "traces" in which or "seeds" in which:
        gen_trace_seeds()
    if "boss" in which:
        gen_boss()
    if "kat" in which:
        gen_kat()
    if "sb3" in which:
        gen_sb3_fixed()
    if "sim" in which:  # after kat: it re-checks the trajectory answers of kat.json on the reference's calculate_score
        gen_sim()


if __name__ == "__main__":
    main()
