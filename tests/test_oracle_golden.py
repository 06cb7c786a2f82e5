"""CPU tests: the C oracle against the committed golden vectors generated from the Python reference
(oracle/gen_golden.py) and against the reference's own known answers (tests/golden/kat.json)."""
import ctypes as C
import json
import os
import zlib
from itertools import combinations

import numpy as np
import pytest

from oracle import pyoracle as po
from tests.helpers import BG_INFO_BEAT_BLIND, BG_INFO_FAILED, GOLD, OBS_KEYS, POLICY_SCRIPTED, TRACES, assert_obs_equal, load_trace, trace_injection


def test_mt_known_answers():
    L = po.lib()
    g = json.load(open(os.path.join(GOLD, "mt_streams.json")))
    for i, seed in enumerate(g["seeds"]):
        mt = po.MT()
        L.bo_mt_seed(C.byref(mt), seed)
        assert [L.bo_mt_u32(C.byref(mt)) for _ in range(16)] == g["u32"][i]
        L.bo_mt_seed(C.byref(mt), seed)
        for n, want in g["randbelow"][i]:
            assert L.bo_mt_randbelow(C.byref(mt), n) == want
        L.bo_mt_seed(C.byref(mt), seed)
        assert [float(L.bo_mt_random(C.byref(mt))).hex() for _ in range(8)] == g["random"][i]


def test_deck_shuffle_and_shop_seed_known_answers():
    """DeterministicRNG stream 0 shuffle (balatro_env_2.py:525) and stream 2 get_int (:1389)."""
    L = po.lib()
    g = json.load(open(os.path.join(GOLD, "mt_streams.json")))
    for i, seed in enumerate(g["seeds"]):
        mt = po.MT()
        L.bo_mt_seed(C.byref(mt), seed % 2 ** 32)
        deck = (C.c_uint8 * 52)(*[(r - 2) * 4 + s for s in range(4) for r in range(2, 15)])
        L.bo_mt_shuffle_u8(C.byref(mt), deck, 52)
        assert list(deck) == g["deck"][i]
        L.bo_mt_seed(C.byref(mt), (seed + 2000) % 2 ** 32)
        assert [L.bo_mt_randbelow(C.byref(mt), 2 ** 31) for _ in range(3)] == g["shop_seed"][i]
    # SURVEY 8(a2) deck KATs
    for seed, first8 in ((42, [36, 41, 49, 12, 33, 50, 13, 3]), (7, [17, 3, 22, 40, 7, 44, 0, 25]),
                         (382, [9, 49, 51, 48, 11, 50, 21, 3])):
        env = po.OracleEnv(seed)
        env.step(45)
        assert env.obs()["hand"].tolist() == first8


def test_classify_golden():
    g = np.load(os.path.join(GOLD, "classify.npz"))
    for cards, n, want in zip(g["cards"], g["n"], g["hand_type"]):
        assert po.classify([int(c) for c in cards[:n]]) == want


def test_classify_all_five_card_hands():
    g = np.load(os.path.join(GOLD, "classify.npz"))
    L = po.lib()
    types = bytearray()
    buf = (C.c_uint8 * 5)()
    for combo in combinations(range(52), 5):
        buf[:] = combo
        types.append(L.bo_classify(buf, 5))
    counts = np.bincount(np.frombuffer(bytes(types), dtype=np.uint8), minlength=12)
    assert counts.tolist() == g["all5_counts"].tolist()
    assert zlib.crc32(bytes(types)) == int(g["all5_crc32"])


@pytest.mark.parametrize("fixture", ["score_hand.json", "score_hand_dups.json"])
def test_score_hand_golden(fixture):
    """UnifiedScorer.score_hand cases generated from the reference; `_dups`: joker lists with repeats (Ankh), several Bloodstones /
    8 Balls drawing per card."""
    cases = json.load(open(os.path.join(GOLD, fixture)))
    L = po.lib()
    for i, c in enumerate(cases):
        cards = [tuple(x) for x in c["cards"]]
        out = po.score_hand(cards, cards[:c["nscoring"]], c["hand_type"], c["style"], c["level"], c["jokers"],
                            c["hands_left"], c["discards_left"], c["deck_len"], c["gseed"])
        ctx = f"case {i}: {c}"
        assert out.score == c["score"], ctx
        assert out.chips == c["chips"] and out.mult == c["mult"], ctx
        assert float(out.x_mult).hex() == c["x_mult"], ctx
        assert out.money == c["money"], ctx
        # the global stream must have advanced by exactly the reference's number of draws
        mt = po.MT()
        L.bo_mt_seed(C.byref(mt), c["gseed"])
        for _ in range(out.draws):
            L.bo_mt_u32(C.byref(mt))
        assert L.bo_mt_u32(C.byref(mt)) == c["probe"], ctx


def test_reference_known_answers():
    """tests/chips_test.py:5-24 and balatro_trajectories.json (the reference's own pinned results)."""
    kat = json.load(open(os.path.join(GOLD, "kat.json")))
    L = po.lib()

    def score(cards, ht):
        chips, mult = C.c_int64(), C.c_int64()
        L.bo_hand_chips_mult.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        L.bo_hand_chips_mult(ht, 1, C.byref(chips), C.byref(mult))
        L.bo_rank_base_chips.restype = C.c_int
        return (chips.value + sum(L.bo_rank_base_chips(r) for r, _ in cards)) * mult.value

    for k in kat["chips_test"]:
        distinct = len({tuple(c) for c in k["cards"]}) == len(k["cards"])
        if distinct:  # five identical cards cannot be classified by balatro_game.py (never emits FLUSH_FIVE)
            assert po.classify([(r - 2) * 4 + s for r, s in k["cards"]]) == k["hand_type"]
        assert score(k["cards"], k["hand_type"]) == k["score"]
    for k in kat["trajectories"]:
        ht = po.classify([(r - 2) * 4 + s for r, s in k["cards"]])
        assert score(k["cards"], ht) == k["score"], k


def test_sim_evaluate_golden():
    """balatro_sim.evaluate_hand (balatro_sim.py:220-366): 10 000 hands with / without Four Fingers / Shortcut -- `top`, and for
    every one of the 12 hand types the number of lists and the first list as positions in the hand."""
    with np.load(os.path.join(GOLD, "sim_eval.npz")) as z:
        g = {k: z[k] for k in z.files}  # (an NpzFile re-inflates the array on every [] access)
    assert (np.bincount(g["e_top"], minlength=12) > 0).all()
    for i in range(len(g["e_n"])):
        cards = [tuple(int(x) for x in c) for c in g["e_cards"][i, :g["e_n"][i]]]
        top, lists = po.sim_evaluate(cards, bool(g["e_ff"][i]), bool(g["e_sc"][i]))
        assert top == g["e_top"][i], (i, cards)
        for t in range(12):
            want = (int(g["e_nlists"][i, t]), [int(p) for p in g["e_pos"][i, t, :g["e_n0"][i, t]]])
            assert lists[t] == want, (i, t, cards, lists[t], want)


def test_sim_score_golden():
    """balatro_sim.calculate_score (balatro_sim.py:402-548): 2 000 cases -- score, money and the position of the global stream
    afterwards -- and the 14 known answers the reference itself holds (balatro_trajectories.json)."""
    with np.load(os.path.join(GOLD, "sim_eval.npz")) as z:
        g = {k: z[k] for k in z.files}
    L = po.lib()
    for i in range(len(g["s_n"])):
        cards = [tuple(int(x) for x in c) for c in g["s_cards"][i, :g["s_n"][i]]]
        jokers = [int(j) for j in g["s_jokers"][i, :g["s_njokers"][i]]]
        o = po.sim_score(cards, jokers, int(g["s_hands_left"][i]), int(g["s_discards_left"][i]), int(g["s_deck_len"][i]), int(g["s_seed"][i]))
        ctx = (i, cards, jokers)
        assert o.score == g["s_score"][i] and o.money == g["s_money"][i], ctx
        mt = po.MT()
        L.bo_mt_seed(C.byref(mt), int(g["s_seed"][i]))
        for _ in range(o.draws):
            L.bo_mt_u32(C.byref(mt))
        assert L.bo_mt_u32(C.byref(mt)) == g["s_probe"][i], ctx
    kat = json.load(open(os.path.join(GOLD, "kat.json")))
    for k in kat["trajectories"]:
        cards = [(r, s, 11 if r == 14 else min(r, 10), 0, 0, 0) for r, s in k["cards"]]
        assert po.sim_score(cards, [], 1, 0, 0, 1).score == k["score"], k


HAND_NAMES = ["High Card", "One Pair", "Two Pair", "Three Kind", "Straight", "Flush", "Full House", "Four Kind", "Straight Flush",
              "Five Kind", "Flush House", "Flush Five"]   # HandType.name.replace('_', ' ').title() (balatro_env_2.py:674)


def replay_trace(name, make_env):
    tr = load_trace(name)
    S, T = tr["actions"].shape
    for si in range(S):
        seed = int(tr["seeds"][si])
        env = make_env(seed, bool(tr["scorer_jokers"]), int(tr["max_ante"]))
        inj = trace_injection(tr, si)

        def inject():
            if inj["jokers"]:
                env.set_jokers(inj["jokers"])
            if inj["money"] is not None:
                env.set_money(inj["money"])
            if inj["ante"] is not None:
                env.set_ante(inj["ante"])
            for (d, e, ed, s) in inj["cards"]:
                env.set_card_state(d, e, ed, s)
            for (ht, l) in inj["levels"]:
                env.set_hand_level(ht, l)
            if inj["consumables"]:
                env.set_consumables(inj["consumables"])

        inject()
        assert_obs_equal(env.obs(), {k: tr["obs0_" + k][si] for k in OBS_KEYS}, f"{name} seed {seed} initial")
        for t in range(T):
            a = int(tr["actions"][si, t])
            if int(tr["policy"]) != POLICY_SCRIPTED:
                assert env.policy_action(int(tr["policy"]), int(tr["policy_seed"]), si, t) == a
            obs, r, term, _, info = env.step(a)
            ctx = f"{name} seed {seed} t {t} action {a}"
            assert r == tr["rewards"][si, t], f"{ctx}: reward {r!r} vs {tr['rewards'][si, t]!r}"
            assert term == bool(tr["terminated"][si, t]), ctx
            assert info.final_score == tr["final_score"][si, t], ctx
            assert info.hand_type == tr["hand_type"][si, t], ctx
            assert info.error == tr["error_code"][si, t], f"{ctx}: error {info.error} vs {tr['error_code'][si, t]} ({tr['error_msg'][si, t]!r})"
            assert info.cards_played == tr["cards_played"][si, t], ctx
            assert bool(info.flags & BG_INFO_BEAT_BLIND) == bool(tr["beat_blind"][si, t]) and bool(info.flags & BG_INFO_FAILED) == bool(tr["failed"][si, t]), ctx
            assert [float(x).hex() for x in info.reward_terms] == [float(x).hex() for x in tr["reward_terms"][si, t]], ctx
            if info.hand_type >= 0:  # info['score_breakdown'] (:909): base / card / joker / final chips, mult, x_mult, money_gained
                bi, bf, bd = tr["breakdown_int"][si, t], tr["breakdown_f64"][si, t], list(info.breakdown)
                want = [bi[5], bi[6], bf[1], bi[2], bi[0], bi[1], bi[7]]
                assert [float(x).hex() for x in bd[:7]] == [float(x).hex() for x in want], f"{ctx}: breakdown {bd} vs {want}"
                assert bi[3] == bi[5] - bi[0] - bi[2] and bi[4] == bi[6] - bi[1] and bf[0] == bf[1], ctx  # joker_* are differences
            msg = str(tr["error_msg"][si, t])
            if info.error == 3:
                assert msg == f"Cannot play {HAND_NAMES[info.aux]} again", (ctx, msg, info.aux)
            elif info.error == 4:
                assert msg == f"Can only play {HAND_NAMES[info.aux]}", (ctx, msg, info.aux)
            elif info.error == 5:
                assert msg == f"Must play at least {info.aux} cards", (ctx, msg, info.aux)
            assert_obs_equal(obs, {k: tr["obs_" + k][si, t] for k in OBS_KEYS}, ctx)
            if term:
                env.reset()
                inject()


def test_forced_boss_coverage():
    """F5: each of the 28 BossBlindTypes is the FIRST boss blind of 24 seeds in both traces (boss_blinds.py:522-532 draws it from the env's
    global stream: the seed picks it), >= 20 accepted plays under every type, every restrictive type rejects at least once
    (can_play_hand, boss_blinds.py:380-407)."""
    for name in ("boss_forced", "boss_forced_scorer"):
        tr = load_trace(name)
        first = tr["obs_boss_blind_type"][:, 0].astype(int)
        assert np.bincount(first, minlength=29)[1:].tolist() == [24] * 28, name
        prev = np.concatenate([tr["obs0_boss_blind_type"][:, None], tr["obs_boss_blind_type"][:, :-1]], axis=1).astype(int)
        prev[:, 1:][tr["terminated"][:, :-1] != 0] = 0
        plays = np.bincount(prev[tr["hand_type"] >= 0], minlength=29)
        assert plays[1:].min() >= 20, (name, plays)
        for code, boss in ((2, 7), (3, 12), (4, 13), (5, 25)):
            assert ((tr["error_code"] == code) & (prev == boss)).sum() >= 1, (name, code)
            assert ((tr["error_code"] == code) & (prev != boss)).sum() == 0, (name, code)


@pytest.mark.parametrize("name", TRACES)
def test_env_trace_golden(name):
    replay_trace(name, lambda seed, scorer, max_ante: po.OracleEnv(seed, scorer_jokers=scorer, max_ante=max_ante))
