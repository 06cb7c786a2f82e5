"""CPU tests that run the imported Python reference side by side with the C oracle.

Only runs where /root/reference exists (the build container); skipped on the GPU box.  This is the direct pin of
the oracle: same seeds, same counter-hash policy, every observation key / reward / info compared bit-exactly.
"""
import random

import numpy as np
import pytest

from oracle import pyoracle as po
from oracle import refharness as rh
from tests.helpers import OBS_KEYS, assert_obs_equal

pytestmark = pytest.mark.skipif(not rh.reference_available(), reason="reference tree not present")

TERMS = ["progress", "milestone", "score", "hand_quality", "efficiency", "synergy", "strategy", "ante_bonus"]


def lockstep(seed, steps, policy, scorer=False, jokers=None, max_ante=0, env_index=0, money=None, ante=None,
             cards=None, levels=None, pseed=11, consumables=None):
    r = rh.RefEnv(seed, scorer_jokers=scorer, max_ante=max_ante)
    o = po.OracleEnv(seed, scorer_jokers=scorer, max_ante=max_ante)

    def inject():
        for e in (r, o):
            if jokers:
                e.set_jokers(jokers)
            if money is not None:
                e.set_money(money)
            if ante is not None:
                e.set_ante(ante)
            for (d, en, ed, s) in (cards or []):
                e.set_card_state(d, en, ed, s)
            for (ht, l) in (levels or []):
                e.set_hand_level(ht, l)
            if consumables:
                e.set_consumables(consumables)

    inject()
    obs_r = r.obs()
    assert_obs_equal(o.obs(), obs_r, f"seed {seed} initial")
    for t in range(steps):
        a = rh.policy_action(obs_r["action_mask"], int(obs_r["phase"]), policy, pseed, env_index, t)
        assert o.policy_action(policy, pseed, env_index, t) == a
        obs_r, rr, tr, _, ir = r.step(a)
        obs_o, ro, to, _, io = o.step(a)
        ctx = f"seed {seed} t {t} action {a}"
        assert_obs_equal(obs_o, obs_r, ctx)
        assert ro == rr and to == tr, f"{ctx}: reward {ro!r} vs {rr!r}"
        if "final_score" in ir:
            assert io.final_score == ir["final_score"] and io.hand_type == int(ir["hand_type"]), ctx
            assert io.cards_played == ir["cards_played"], ctx
            for i, k in enumerate(TERMS):
                assert io.reward_terms[i] == float(ir["reward_breakdown"][k]), f"{ctx}: {k}"
        if tr:
            r.reset()
            o.reset()
            inject()
            obs_r = r.obs()
            assert_obs_equal(o.obs(), obs_r, ctx + " reset")


@pytest.mark.parametrize("policy", [rh.POLICY_SMALL_ONLY, rh.POLICY_CYCLE3, rh.POLICY_UNIFORM])
def test_env_lockstep_no_jokers(policy):
    for s in range(12):
        lockstep(7000 + s, 500, policy, env_index=s)


def test_env_lockstep_scorer_jokers():
    from oracle.gen_golden import IMPLEMENTED
    for s in range(12):
        lockstep(8000 + s, 500, rh.POLICY_CYCLE3, scorer=True, jokers=random.Random(s).sample(IMPLEMENTED, 5),
                 max_ante=4, env_index=s)


def test_env_lockstep_rich_shop_and_late_antes():
    for s in range(10):
        lockstep(9000 + s, 600, rh.POLICY_UNIFORM, scorer=bool(s & 1), env_index=s, max_ante=22,
                 money=[500, 3000, 100000][s % 3], ante=[1, 2, 5, 9, 20][s % 5],
                 jokers=random.Random(s).sample(range(1, 151), s % 6))


def test_env_lockstep_card_states_and_levels():
    for s in range(10):
        rr = random.Random(40 + s)
        cards = [(d, rr.choice([0, 0, 1, 2, 3, 4, 5, 6, 7, 8]), rr.choice([0, 0, 1, 2, 3]), rr.choice([0, 0, 1, 2, 3]))
                 for d in range(12)]
        levels = [(ht, rr.randint(1, 15)) for ht in range(9)]
        lockstep(9500 + s, 400, rh.POLICY_UNIFORM, scorer=bool(s & 1), env_index=s, max_ante=20, cards=cards,
                 levels=levels, jokers=rr.sample(range(1, 151), s % 6))


def test_env_lockstep_consumables():
    """Tarot / spectral / planet use (balatro_env_2.py:1066-1172, consumables.py) incl. the cases where the reference
    raises (harness convention: reward -1.0, state as the exception left it), and Immolate / Cryptid (deck length changes)."""
    pool = list(range(1, 23)) + list(range(30, 42)) + list(range(50, 68))
    for s in range(26):
        rr = random.Random(70 + s)
        cards = [(d, rr.choice([0, 0, 4, 8]), 0, rr.choice([0, 4])) for d in range(16)] if s % 2 else None
        lockstep(9700 + s, 400, rh.POLICY_UNIFORM, scorer=bool(s & 1), env_index=s, max_ante=20, cards=cards,
                 consumables=[pool[(2 * s) % len(pool)], pool[(2 * s + 1) % len(pool)]], money=[None, 30][s % 2],
                 jokers=([53] if s % 3 == 0 else []) + rr.sample(range(1, 53), s % 5))


def test_env_lockstep_immolate_cryptid():
    """Immolate (five sampled cards leave the live deck list: every later index shifts) and Cryptid (two foreign copies
    appended) over and over, with The Fool copying them and Blue Joker reading len(deck)."""
    for s in range(16):
        lockstep(9800 + s, 500, rh.POLICY_UNIFORM, scorer=bool(s & 1), env_index=s, max_ante=20,
                 consumables=[[59, 65], [65, 59], [59, 1], [65, 1], [59, 59], [1, 59]][s % 6], jokers=[53, 1][: 1 + s % 2],
                 cards=[(d, [0, 7, 4][d % 3], 0, [0, 4][d % 2]) for d in range(20)] if s % 4 == 0 else None)


def test_env_lockstep_forced_rare_hands():
    """Decks arranged so that the first play IS a chosen hand type (straight flush, four of a kind, flush ... at
    deck[0..k-1]; classification reads deck[position]), at antes 1..8, small / big / boss blinds, with and without
    scorer-level jokers: reference and oracle in lockstep through the play and 40 more steps."""
    from tests.helpers import forced_deck, forced_hand_script
    rr = random.Random(5)
    seen = [0] * 9
    for s in range(9 * 12):
        ht, scorer = s % 9, bool((s // 9) & 1)
        seed = 9900 + s
        r = rh.RefEnv(seed, scorer_jokers=scorer)
        o = po.OracleEnv(seed, scorer_jokers=scorer)
        deck, k = forced_deck(ht, rr)
        jokers = rr.sample(range(1, 151), rr.randint(0, 5)) if scorer else [113, 40, 33][: s % 4]
        ante = 1 + (s // 18) % 8
        for e in (r, o):
            e.set_jokers(jokers); e.set_ante(ante); e.set_deck(deck)
        script = forced_hand_script(k, rr, blind=[45, 46, 47][(s // 9) % 3])
        obs_r = r.obs()
        assert_obs_equal(o.obs(), obs_r, f"forced {s} initial")
        for t in range(len(script) + 40):
            a = script[t] if t < len(script) else rh.policy_action(obs_r["action_mask"], int(obs_r["phase"]), 0, 31, s, t)
            obs_r, rr_, tr, _, ir = r.step(a)
            obs_o, ro, to, _, io = o.step(a)
            ctx = f"forced {s} ht {ht} t {t} action {a}"
            assert_obs_equal(obs_o, obs_r, ctx)
            assert ro == rr_ and to == tr, f"{ctx}: reward {ro!r} vs {rr_!r}"
            if "final_score" in ir:
                assert io.final_score == ir["final_score"] and io.hand_type == int(ir["hand_type"]), ctx
                for i, kk in enumerate(TERMS):
                    assert io.reward_terms[i] == float(ir["reward_breakdown"][kk]), f"{ctx}: {kk}"
                if t == len(script) - 1:
                    assert io.hand_type == ht, ctx
                    seen[ht] += 1
            if tr:
                break
    assert min(seen) >= 6, seen


def test_sim_evaluate_and_score_lockstep():
    """balatro_sim.BalatroSimulator (imported with the SURVEY App. C shim) against oracle/bo_sim.c on fresh random hands."""
    import ctypes as C
    from oracle.gen_golden import IMPLEMENTED, sim_random_hand
    r = random.Random(99)
    for i in range(1500):
        cards = sim_random_hand(r)
        ff, sc = bool(i & 1), bool(i & 2)
        assert rh.sim_evaluate(cards, ff, sc) == po.sim_evaluate(cards, ff, sc), (cards, ff, sc)
    L = po.lib()
    for i in range(600):
        cards = sim_random_hand(r, n=r.choice([1, 2, 3, 4, 5, 5, 5]))
        jokers = r.sample(IMPLEMENTED + [18, 69, 100], r.randint(0, 5))
        gs = None if i % 2 else {"hands_left": r.randint(1, 4), "discards_left": r.randint(0, 3)}
        dl, seed = r.choice([0, 40, 52]), r.randrange(2 ** 32)
        score, money, probe = rh.sim_score(cards, jokers, gs, dl, seed)
        o = po.sim_score(cards, jokers, 1 if gs is None else gs["hands_left"], 0 if gs is None else gs["discards_left"], dl, seed)
        mt = po.MT()
        L.bo_mt_seed(C.byref(mt), seed)
        for _ in range(o.draws):
            L.bo_mt_u32(C.byref(mt))
        assert (o.score, o.money, L.bo_mt_u32(C.byref(mt))) == (score, money, probe), (cards, jokers, gs, dl, seed)


def test_reseed_reproduces_first_shuffle():
    """reset(seed=s) rebuilds the streams (balatro_env_2.py:507-509); reset() continues them."""
    r = rh.RefEnv(42)
    o = po.OracleEnv(42)
    for e in (r, o):
        e.step(45)
    first = r.obs()["hand"].tolist()
    assert o.obs()["hand"].tolist() == first
    r.reset(); o.reset()
    for e in (r, o):
        e.step(45)
    second = r.obs()["hand"].tolist()
    assert second != first and o.obs()["hand"].tolist() == second
    r.reset(seed=42); o.reset(seed=42)
    for e in (r, o):
        e.step(45)
    assert r.obs()["hand"].tolist() == first and o.obs()["hand"].tolist() == first


def test_consumable_deck_fence_placement():
    """WHERE the `BG_ERR_CONSUMABLE_DECK` fence stands relative to the reference.
    Immolate (consumables.py:519-531) removes five sampled cards from the live deck list.  The restatement follows it while the hand's deck indexes
    (always 0..7) stay valid behind the use, i.e. while 13 or more real cards are left -- EIGHT uses in one episode, 52 -> 12 cards, in lockstep with the
    reference, the last two sampled from a pool (random.sample's method for n <= 21) -- and REFUSES the ninth (reward -1.0, error 12, state untouched).
    The reference accepts it (reward 7.0) and plays on with a deck of 7, 2, 0 cards while its hand still holds indexes up to 7: from there on its
    unguarded deck[i] reads (balatro_env_2.py:577,670,937) raise IndexError as soon as a play or a boss blind touches one -- the first step after the
    deck is empty at the latest.  So the fence now stands where the reference's own state stops being one it can step from."""
    r = rh.RefEnv(4242, scorer_jokers=False, max_ante=20)
    o = po.OracleEnv(4242, scorer_jokers=False, max_ante=20)
    for e in (r, o):
        e.step(45)
    uses_followed = 0
    for it in range(9):
        for e in (r, o):
            e.set_consumables([59])   # Immolate
        for act in (2, 3, 10):
            obs_r, rr, tr, _, ir = r.step(act)
            obs_o, ro, to, _, io = o.step(act)
            if act != 10:
                assert_obs_equal(obs_o, obs_r, f"use {it} action {act}")
                continue
            if it < 8:
                assert_obs_equal(obs_o, obs_r, f"Immolate {it}")
                assert ro == rr == 7.0 and int(obs_r["deck_size"]) == 47 - 5 * it
                uses_followed += 1
            else:   # the fence: the reference goes on, the restatement refuses
                assert rr == 7.0 and len(r.env.state.deck) == 7
                assert ro == -1.0 and io.error == 12 and int(obs_o["deck_size"]) == 12
    assert uses_followed == 8
    # ... and the reference alone: down to an empty deck, then IndexError on the next play
    for it in range(2):
        r.set_consumables([59])
        for act in (2, 3, 10):
            r.step(act)
    assert len(r.env.state.deck) == 0
    with pytest.raises(IndexError):
        for act in (2, 3, 0):
            r.step(act)


def test_immolate_pool_with_cryptid_copies_vs_reference():
    """random.sample's POOL method (n <= 21) over a deck that ends in Cryptid's copies (consumables.Card objects behind the real cards): two Cryptids,
    then Immolates down to 12 real cards -- the last ones draw from a pool whose tail are copies (a picked copy leaves through deck.remove of an equal
    copy, a picked real card shifts every later index) -- in lockstep with the reference, observation by observation."""
    r = rh.RefEnv(5151, scorer_jokers=False, max_ante=20)
    o = po.OracleEnv(5151, scorer_jokers=False, max_ante=20)
    for e in (r, o):
        e.step(45)
    def counts():
        real = sum(1 for c in r.env.state.deck if type(c).__module__.endswith("cards"))
        return real, len(r.env.state.deck)
    pool_with_copies, uses = 0, 0
    for it, cid in enumerate([65, 65] + [59] * 10):
        real, total = counts()
        if cid == 59 and real < 13:
            break
        pool_with_copies += cid == 59 and total <= 21 and total > real
        for e in (r, o):
            e.set_consumables([cid])
        for act in (2, 3, 10):
            obs_r, rr, tr, _, ir = r.step(act)
            obs_o, ro, to, _, io = o.step(act)
            assert_obs_equal(obs_o, obs_r, f"use {it} (id {cid}) action {act}")
            assert ro == rr, (it, act, ro, rr)
        uses += 1
    assert pool_with_copies >= 1 and uses >= 10, (pool_with_copies, uses)   # the pool method really ran with copies in the pool


def test_cryptid_int8_fence_placement():
    """Cryptid (consumables.py:581-591) appends two copies per use; `deck_size` is np.int8(len(deck)) (balatro_env_2.py:1491).  The restatement follows the
    reference up to 126 cards (37 uses on a full deck) and refuses the use that would make 128 -- where the reference's OWN observation stops being
    defined: np.int8(128) raises OverflowError under numpy >= 2 and wraps to -128 under numpy 1 (it was fenced at 60 copies before)."""
    r = rh.RefEnv(6262, scorer_jokers=False, max_ante=20)
    o = po.OracleEnv(6262, scorer_jokers=False, max_ante=20)
    for e in (r, o):
        e.step(45)
    for it in range(37):
        for e in (r, o):
            e.set_consumables([65])
        for act in (2, 3, 10):
            obs_r, rr, tr, _, ir = r.step(act)
            obs_o, ro, to, _, io = o.step(act)
            assert_obs_equal(obs_o, obs_r, f"Cryptid {it} action {act}")
            assert ro == rr, (it, act)
    assert int(obs_o["deck_size"]) == 126 and len(r.env.state.deck) == 126
    o.set_consumables([65]); r.set_consumables([65])
    for act in (2, 3):
        o.step(act); r.step(act)
    obs_o, ro, to, _, io = o.step(10)
    assert ro == -1.0 and io.error == 12 and int(obs_o["deck_size"]) == 126
    try:
        obs_r, rr, *_ = r.step(10)
        assert int(obs_r["deck_size"]) == -128 and len(r.env.state.deck) == 128      # numpy 1: wrapped
    except OverflowError:
        assert len(r.env.state.deck) == 128                                            # numpy >= 2: the reference's step raises out of _get_observation


def test_reference_wrapper_over_the_drop_in_space():
    """The reference's own `BalatroEnvFixed.__init__` (train_balatro_fixed.py:20-124) run over the DROP-IN's observation space instead
    of the reference env's: it walks `self.env.observation_space.spaces.items()`, so it must come out with the same 51-key fixed
    space it builds over `balatro_env_2.BalatroEnv` (the space recorded in tests/golden/sb3_fixed.npz) -- key order, dtypes, shapes and
    the bounds of every Box it keeps as is.  No GPU: a stand-in env class that carries only the two spaces takes the drop-in's place."""
    import contextlib
    import importlib
    import io
    import os
    tbf = rh.load_fixed_wrappers()          # (installs the gymnasium stand-in the reference is imported with)
    import balatro_gym_amd.env as envmod
    envmod = importlib.reload(envmod)        # the drop-in builds its space from whatever `gymnasium` is importable: now the stand-in
    space = envmod.make_observation_space()

    class SpacesOnly:
        def __init__(self, seed=None):
            self.observation_space = space
            self.action_space = envmod._spaces.Discrete(60)

    ref_cls = tbf.OriginalBalatroEnv
    try:
        tbf.OriginalBalatroEnv = SpacesOnly
        with contextlib.redirect_stdout(io.StringIO()):
            over_drop_in = tbf.BalatroEnvFixed(seed=1)
        tbf.OriginalBalatroEnv = ref_cls
        with contextlib.redirect_stdout(io.StringIO()):
            over_reference = tbf.BalatroEnvFixed(seed=1)
    finally:
        tbf.OriginalBalatroEnv = ref_cls
    a, b = over_drop_in.observation_space.spaces, over_reference.observation_space.spaces
    assert list(a) == list(b) and len(a) == 51
    assert over_drop_in.space_transforms == over_reference.space_transforms
    for k in a:
        assert a[k].dtype == b[k].dtype and tuple(a[k].shape) == tuple(b[k].shape), k
        assert np.array_equal(np.asarray(a[k].low, dtype=np.float64), np.asarray(b[k].low, dtype=np.float64)), k
        assert np.array_equal(np.asarray(a[k].high, dtype=np.float64), np.asarray(b[k].high, dtype=np.float64)), k
    from tests.helpers import GOLD
    with np.load(os.path.join(GOLD, "sb3_fixed.npz")) as z:
        assert [str(x) for x in z["keys"]] == list(a)
        assert [str(x) for x in z["dtypes"]] == [a[k].dtype.name for k in a]
