
"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI (ctypes ->
libbalatro_mi355x.so), against (1) the committed golden traces generated from the Python reference and (2) the CPU
oracle on fresh seeds.  Integer / byte / index results and float64 rewards must be BIT-exact (tolerance 0).
"""
import os
import random

import numpy as np
import pytest

from tests.helpers import OBS_KEYS, load_trace, trace_injection

HAND_TYPE_NAMES = ["High Card", "One Pair", "Two Pair", "Three Kind", "Straight", "Flush", "Full House", "Four Kind", "Straight Flush",
                   "Five Kind", "Flush House", "Flush Five"]

pytestmark = pytest.mark.gpu

# BG_TEST_SEED_OFFSET=<k> moves every oracle-compared test below to other games (a soak runs the suite under several offsets)
SEED_OFFSET = int(os.environ.get("BG_TEST_SEED_OFFSET", "0"))

GPU_TRACES = ["c1_small_only", "c2_cycle3", "c3_jokers", "c5_uniform", "c5_uniform_rich", "cards_levels", "seeds_special", "consumables",
              "consumables_scorer", "boss_forced", "boss_forced_scorer"]
TERMS = 8


def _vec(n, seeds, **kw):
    from balatro_gym_amd import BalatroVecEnv
    return BalatroVecEnv(n, seeds, device=0, **kw)


def _obs_np(env):
    return {k: v.cpu().numpy() for k, v in env.obs.items()}


def _assert_obs(got, want_rows, ctx):
    for k in OBS_KEYS:
        g, w = got[k], want_rows[k]
        if not np.array_equal(g, w):
            bad = np.argwhere(g.reshape(len(g), -1) != w.reshape(len(w), -1))[0][0]
            raise AssertionError(f"{ctx}: obs[{k}] differs for env {bad}: got {g[bad]} want {w[bad]}")


@pytest.mark.parametrize("name", GPU_TRACES)
def test_golden_trace(name):
    """Replay a reference trace: all of its seeds in parallel, same actions, compare everything every step."""
    import torch
    tr = load_trace(name)
    S, T = tr["actions"].shape
    seeds = [int(s) for s in tr["seeds"]]
    inj = [trace_injection(tr, si) for si in range(S)]
    has_cons = any(i["consumables"] for i in inj)
    has_cards = any(i["cards"] for i in inj) or has_cons  # tarot / spectral cards edit card states
    env = _vec(S, seeds, scorer_jokers=bool(tr["scorer_jokers"]), autoreset=False, max_ante=int(tr["max_ante"]),
               card_states=has_cards)
    if has_cards:  # cards.py CardState per deck index, re-applied after every reset like the other injections
        env.inject_cards([i["cards"] for i in inj], apply_now=True)
    if any(i["jokers"] or i["money"] is not None or i["ante"] is not None or i["levels"] for i in inj):
        levels = np.zeros((S, 12), np.uint8)
        for si, i in enumerate(inj):
            for ht, l in i["levels"]:
                levels[si, ht] = l
        env.inject(jokers=[i["jokers"] for i in inj],
                   money=[-1 if i["money"] is None else i["money"] for i in inj],
                   ante=[0 if i["ante"] is None else i["ante"] for i in inj],
                   levels=levels if levels.any() else None, apply_now=True)
        env.observe()
    if has_cons:  # state.consumables by id, re-applied after every reset
        env.inject_consumables([i["consumables"] for i in inj], apply_now=True)
    _assert_obs(_obs_np(env), {k: tr["obs0_" + k] for k in OBS_KEYS}, f"{name} initial")
    for t in range(T):
        a = torch.from_numpy(tr["actions"][:, t].astype(np.int32)).to(env.device)
        _, reward, term, trunc, info = env.step(a)
        ctx = f"{name} t {t}"
        r = reward.cpu().numpy()
        assert np.array_equal(r.view(np.uint64), tr["rewards"][:, t].view(np.uint64)), \
            f"{ctx}: rewards {r} vs {tr['rewards'][:, t]}"
        tm = term.cpu().numpy()
        assert np.array_equal(tm, tr["terminated"][:, t]), ctx
        assert not trunc.cpu().numpy().any()
        assert np.array_equal(info["final_score"].cpu().numpy(), tr["final_score"][:, t]), ctx
        assert np.array_equal(info["hand_type"].cpu().numpy(), tr["hand_type"][:, t]), ctx
        err = info["error"].cpu().numpy()
        assert np.array_equal(err, tr["error_code"][:, t].astype(np.int32)), f"{ctx}: error codes {err} vs {tr['error_code'][:, t]}"  # the exact BG_ERR_* (11: the reference raised)
        assert np.array_equal(info["cards_played"].cpu().numpy(), tr["cards_played"][:, t]), ctx
        fl = info["flags"].cpu().numpy()
        assert np.array_equal((fl & 1) != 0, tr["beat_blind"][:, t] != 0) and np.array_equal((fl & 2) != 0, tr["failed"][:, t] != 0), ctx
        assert np.array_equal(info["reward_terms"].cpu().numpy().view(np.uint64), tr["reward_terms"][:, t].view(np.uint64)), ctx
        # info['score_breakdown'] (:909) of the accepted plays, and what the boss blinds' rejection messages name (boss_blinds.py:393-405)
        bd = info["score_breakdown"].cpu().numpy()
        bi, bf = tr["breakdown_int"][:, t], tr["breakdown_f64"][:, t]
        want = np.stack([bi[:, 5], bi[:, 6], bf[:, 1], bi[:, 2], bi[:, 0], bi[:, 1], bi[:, 7]], axis=1).astype(np.float64)
        played = tr["hand_type"][:, t] >= 0
        assert np.array_equal(bd[played, :7].view(np.uint64), want[played].view(np.uint64)), f"{ctx}: score_breakdown"
        aux = info["aux"].cpu().numpy()
        for si in np.nonzero((err >= 3) & (err <= 5))[0]:
            msg = str(tr["error_msg"][si, t])
            wantmsg = (f"Cannot play {HAND_TYPE_NAMES[aux[si]]} again" if err[si] == 3 else
                       f"Can only play {HAND_TYPE_NAMES[aux[si]]}" if err[si] == 4 else f"Must play at least {aux[si]} cards")
            assert msg == wantmsg, (ctx, si, msg, wantmsg)
        _assert_obs(_obs_np(env), {k: tr["obs_" + k][:, t] for k in OBS_KEYS}, ctx)
        if tm.any():
            env.reset(mask=torch.from_numpy(tm).to(env.device))  # un-seeded reset() + reset template
    env.check()
    env.close()


def _oracle_envs(n, seeds, scorer, max_ante, jokers=None):
    from oracle import pyoracle as po
    envs = [po.OracleEnv(seeds[i], scorer_jokers=scorer, max_ante=max_ante) for i in range(n)]
    if jokers:
        for e, js in zip(envs, jokers):
            e.set_jokers(js)
    return envs


@pytest.mark.parametrize("policy,scorer", [(0, False), (2, False), (2, True), (0, True)])
def test_step_vs_oracle_fresh_seeds(policy, scorer):
    """512 fresh seeds x 250 steps in lockstep with the CPU oracle (actions from the oracle's counter-hash policy)."""
    import torch
    from oracle import pyoracle as po
    from oracle.gen_golden import IMPLEMENTED
    n, T = 512, 250
    seeds = [910_000 + SEED_OFFSET + 7 * i for i in range(n)]
    jokers = [random.Random(i).sample(IMPLEMENTED if i % 2 else list(range(1, 151)), i % 6) for i in range(n)] if scorer else None
    max_ante = 4 if scorer else 0
    env = _vec(n, seeds, scorer_jokers=scorer, autoreset=False, max_ante=max_ante)
    if jokers:
        env.inject(jokers=jokers, apply_now=True)
        env.observe()
    orc = _oracle_envs(n, seeds, scorer, max_ante, jokers)
    want0 = [o.obs() for o in orc]
    _assert_obs(_obs_np(env), {k: np.stack([w[k] for w in want0]) for k in OBS_KEYS}, "initial")
    for t in range(T):
        acts = np.array([o.policy_action(policy, 99, i, t) for i, o in enumerate(orc)], dtype=np.int32)
        res = [o.step(int(a)) for o, a in zip(orc, acts)]
        _, reward, term, _, info = env.step(torch.from_numpy(acts).to(env.device))
        ctx = f"policy {policy} scorer {scorer} t {t}"
        wr = np.array([r[1] for r in res])
        assert np.array_equal(reward.cpu().numpy().view(np.uint64), wr.view(np.uint64)), ctx
        wt = np.array([r[2] for r in res], dtype=np.uint8)
        assert np.array_equal(term.cpu().numpy(), wt), ctx
        assert np.array_equal(info["final_score"].cpu().numpy(), np.array([r[4].final_score for r in res])), ctx
        assert np.array_equal(info["hand_type"].cpu().numpy(), np.array([r[4].hand_type for r in res], dtype=np.int8)), ctx
        assert np.array_equal(info["cards_played"].cpu().numpy(), np.array([r[4].cards_played for r in res], dtype=np.int8)), ctx
        assert np.array_equal(info["error"].cpu().numpy(), np.array([r[4].error for r in res], dtype=np.int32)), ctx
        assert np.array_equal(info["flags"].cpu().numpy() & 511, np.array([r[4].flags for r in res], dtype=np.int32)), ctx
        wterms = np.array([[r[4].reward_terms[i] for i in range(TERMS)] for r in res])
        assert np.array_equal(info["reward_terms"].cpu().numpy().view(np.uint64), wterms.view(np.uint64)), ctx
        _assert_obs(_obs_np(env), {k: np.stack([r[0][k] for r in res]) for k in OBS_KEYS}, ctx)
        if wt.any():
            for i in np.nonzero(wt)[0]:
                orc[i].reset()
                if jokers:
                    orc[i].set_jokers(jokers[i])
            env.reset(mask=torch.from_numpy(wt).to(env.device))
    env.check()
    env.close()


@pytest.mark.parametrize("many", [0, 40])
def test_step_path_across_refill_periods_vs_oracle(many):
    """900 steps through bg_step (one launch per step) and through bg_step_many (40 steps per call): more than two refill periods, so the look-ahead refill
    runs BESIDE the step launches -- its scan behind the launch that asks for it, its dense kernels in pieces behind the following ones (round 5; a
    synchronous refill on the stream before) -- with a masked reset and a state blob in between.  Observation, reward, termination and info of every
    step against the oracle."""
    import torch
    from oracle.gen_golden import IMPLEMENTED
    n, T = 160, 900
    seeds = [915_000 + SEED_OFFSET + 3 * i for i in range(n)]
    jokers = [random.Random(8000 + i).sample(IMPLEMENTED, i % 6) for i in range(n)]
    env = _vec(n, seeds, scorer_jokers=True, autoreset=False, max_ante=4)
    env.inject(jokers=jokers, apply_now=True)
    env.observe()
    orc = _oracle_envs(n, seeds, True, 4, jokers)
    if many:
        from balatro_gym_amd.vec_env import ObsBuffers
        ob_k = ObsBuffers(n, env.device, steps=many)
        reward_k = torch.zeros((many, n), dtype=torch.float64, device=env.device)
        term_k = torch.zeros((many, n), dtype=torch.uint8, device=env.device)
    t = 0
    while t < T:
        k = min(many, T - t) if many else 1
        acts = np.zeros((k, n), np.int32)
        res_k = []
        for j in range(k):
            acts[j] = [o.policy_action(2, 77, i, t + j) for i, o in enumerate(orc)]
            res = [o.step(int(a)) for o, a in zip(orc, acts[j])]
            res_k.append(res)
            for i, r in enumerate(res):   # autoreset off: the oracle side resets what terminated (the device side below, by mask, when stepping one by one)
                if r[2]:
                    orc[i].reset(); orc[i].set_jokers(jokers[i])
            if many and any(r[2] for r in res):   # step_many stops nowhere: end the block at a termination so that both sides reset here
                acts = acts[:j + 1]; k = j + 1
                break
        if many:
            env.step_many(torch.from_numpy(acts).to(env.device), obs_buffers=ob_k, reward=reward_k, terminated=term_k)
            wr = np.array([[r[1] for r in res] for res in res_k])
            assert np.array_equal(reward_k[:k].cpu().numpy().view(np.uint64), wr.view(np.uint64)), f"t {t}"
            wt = np.array([[r[2] for r in res] for res in res_k], dtype=np.uint8)
            assert np.array_equal(term_k[:k].cpu().numpy(), wt), f"t {t}"
            for j in (0, k - 1):
                _assert_obs({key: ob_k.tensors[key][j].cpu().numpy() for key in OBS_KEYS}, {key: np.stack([r[0][key] for r in res_k[j]]) for key in OBS_KEYS}, f"t {t} + {j}")
            tm = wt[-1]
        else:
            _, reward, term, _, info = env.step(torch.from_numpy(acts[0]).to(env.device))
            last = res_k[0]
            assert np.array_equal(reward.cpu().numpy().view(np.uint64), np.array([r[1] for r in last]).view(np.uint64)), f"t {t}"
            tm = np.array([r[2] for r in last], dtype=np.uint8)
            assert np.array_equal(term.cpu().numpy(), tm), f"t {t}"
            assert np.array_equal(info["error"].cpu().numpy(), np.array([r[4].error for r in last], dtype=np.int32)), f"t {t}"
            _assert_obs(_obs_np(env), {key: np.stack([r[0][key] for r in last]) for key in OBS_KEYS}, f"t {t}")
        if tm.any():
            env.reset(mask=torch.from_numpy(tm).to(env.device))
        t += k
        if t in (401, 440):
            env.get_state(3)
    env.observe()
    _assert_obs(_obs_np(env), {key: np.stack([o.obs()[key] for o in orc]) for key in OBS_KEYS}, "final")
    env.check()
    env.close()


def test_shop_stream_beyond_slot_vs_oracle():
    """A shop visit with dozens of rerolls reads far more of `random.Random(shop_seed)` than a ring slot keeps (words
    0..131 and 396..527 of the seeded state): the stream is re-seeded in full into the overflow block and the visit carries
    on there, through the two- and three-level regeneration of words beyond 227.  Every step against the oracle."""
    import torch
    n, T = 96, 260
    seeds = [77_000 + SEED_OFFSET + 3 * i for i in range(n)]
    env = _vec(n, seeds, autoreset=False)
    env.inject(money=[10_000_000] * n, apply_now=True)
    env.observe()
    orc = _oracle_envs(n, seeds, False, 0)
    for o in orc:
        o.set_money(10_000_000)
    rerolls = [0] * n
    deepest = 0
    for t in range(T):
        acts = np.zeros(n, np.int32)
        for i, o in enumerate(orc):
            ob = o.obs()
            if int(ob["phase"]) == 1:  # SHOP: reroll up to 10 + i % 36 times per visit, then leave
                if ob["action_mask"][30] and rerolls[i] < 10 + i % 36:
                    acts[i] = 30; rerolls[i] += 1
                else:
                    acts[i] = 31; deepest = max(deepest, rerolls[i]); rerolls[i] = 0
            else:
                acts[i] = o.policy_action(1, 5, i, t)  # small blind, uniform play
        res = [o.step(int(a)) for o, a in zip(orc, acts)]
        _, reward, term, _, info = env.step(torch.from_numpy(acts).to(env.device))
        ctx = f"t {t}"
        assert np.array_equal(reward.cpu().numpy().view(np.uint64), np.array([r[1] for r in res]).view(np.uint64)), ctx
        wt = np.array([r[2] for r in res], dtype=np.uint8)
        assert np.array_equal(term.cpu().numpy(), wt), ctx
        _assert_obs(_obs_np(env), {k: np.stack([r[0][k] for r in res]) for k in OBS_KEYS}, ctx)
        if wt.any():
            for i in np.nonzero(wt)[0]:
                orc[i].reset(); orc[i].set_money(10_000_000); rerolls[i] = 0
            env.reset(mask=torch.from_numpy(wt).to(env.device))
    env.check()
    env.close()
    assert deepest >= 30, deepest  # visits of 30+ inventories (~300+ words) really happened


def _oracle_rollout(n, seeds, T, policy, pseed, scorer, max_ante, jokers, env_index0=0, t0=0, cards=None, consumables=None, action_fn=None,
                    env_indexes=None, caps=None):
    """SAME_STEP auto-reset rollout on the oracle; returns per-step obs/rewards/terminated and the stats dict.  `action_fn(oracle_env, i, t)`
    replaces the counter-hash policy; `env_indexes[i]` = the GLOBAL index env i has in the job it is a slice of (default env_index0 + i);
    `caps` = {step: per-env curriculum caps set BEFORE that step} (CurriculumBalatroEnv.current_max_ante, per env)."""
    orc = _oracle_envs(n, seeds, scorer, max_ante, jokers)
    if cards:
        for o, cs in zip(orc, cards):
            for (idx, e_, d_, s_) in cs:
                o.set_card_state(idx, e_, d_, s_)
    if consumables:
        for o, cs in zip(orc, consumables):
            o.set_consumables(cs)
    obs = {k: [] for k in OBS_KEYS}
    rewards = np.zeros((T, n)); terms = np.zeros((T, n), np.uint8); acts = np.zeros((T, n), np.int32)
    stats = {"steps": 0, "episodes": 0, "plays": 0, "score_sum": 0, "reward_bits": 0}
    for t in range(T):
        row = []
        if caps and t in caps:
            for o, c in zip(orc, caps[t]):
                o.set_max_ante(int(c))
        for i, o in enumerate(orc):
            a = action_fn(o, i, t) if action_fn else o.policy_action(policy, pseed, env_indexes[i] if env_indexes is not None else env_index0 + i, t0 + t)
            ob, r, term, _, info = o.step(a)
            if term:
                ob = o.reset()
                if jokers:
                    o.set_jokers(jokers[i])
                    ob = o.obs()
                if cards:
                    for (idx, e_, d_, s_) in cards[i]:
                        o.set_card_state(idx, e_, d_, s_)
                if consumables:
                    o.set_consumables(consumables[i])
                    ob = o.obs()
                stats["episodes"] += 1
            rewards[t, i] = r; terms[t, i] = term; acts[t, i] = a
            stats["steps"] += 1
            stats["reward_bits"] ^= (int(np.float64(r).view(np.uint64)) * (2 * (t0 + t) + 1)) & (2 ** 64 - 1)
            if info.hand_type >= 0:
                stats["plays"] += 1
                stats["score_sum"] += info.final_score
            row.append(ob)
        for k in OBS_KEYS:
            obs[k].append(np.stack([r[k] for r in row]))
    return {k: np.stack(v) for k, v in obs.items()}, rewards, terms, acts, stats


@pytest.mark.parametrize("layout", ["keys", "rows"])
def test_replayed_rollout_actions_reproduce_the_rollout(layout):
    """What bench.py's `step_path` block does: the actions of a fused rollout, recorded per step, replayed through bg_step / bg_step_rows on a twin handle
    -- every step's reward and termination flag and the final observation must be the rollout's (both sides are pinned to the oracle elsewhere; this pins
    the RECORDING: it needs per-step buffers, the wrapper refuses a [T, N] tensor without them), and the replay must contain service steps."""
    import torch
    from balatro_gym_amd.vec_env import ObsBuffers
    from oracle.gen_golden import IMPLEMENTED
    n, T = 512, 120
    seeds = [61_000 + SEED_OFFSET + i for i in range(n)]
    jokers = [random.Random(300 + i).sample(IMPLEMENTED, 5) for i in range(n)]

    def make(**kw):
        e = _vec(n, seeds, scorer_jokers=True, autoreset=True, max_ante=4, **kw)
        e.inject(jokers=jokers, apply_now=True)
        return e
    a = make()
    ob = ObsBuffers(n, a.device, steps=T)
    reward = torch.zeros((T, n), dtype=torch.float64, device=a.device)
    term = torch.zeros((T, n), dtype=torch.uint8, device=a.device)
    acts = torch.zeros((T, n), dtype=torch.int32, device=a.device)
    with pytest.raises(ValueError, match="row 0"):
        a.rollout(T, policy=2, policy_seed=99, actions=acts)
    a.rollout(T, policy=2, policy_seed=99, obs_buffers=ob, reward=reward, terminated=term, actions=acts)
    plays = a.stats()["plays"]
    assert plays > n and int((acts == 0).sum().item()) >= plays          # PLAY_HAND (action 0) is in every row, not only in row 0
    assert all(int((acts[t] != 0).sum().item()) > 0 for t in (1, T // 2, T - 1))
    b = make(obs_layout=layout)
    for t in range(T):
        _, r, tm, _, _ = b.step(acts[t])
        assert torch.equal(r.view(torch.int64), reward[t].view(torch.int64)), f"reward of step {t}"
        assert torch.equal(tm.to(torch.uint8), term[t]), f"terminated of step {t}"
    b.check()
    got = _obs_np(b)
    for k in OBS_KEYS:
        assert np.array_equal(got[k], ob.tensors[k][T - 1].cpu().numpy()), k
    # ... and the launches that overwrite the observation arrays in place (the library writes them ONCE, behind the last step: bg_engine.h keys_at_end):
    # all T steps as one bg_step_many call, and the same rollout again without per-step buffers -- the live observation is the last step's
    c = make(obs_layout=layout)
    _, r, tm, _, _ = c.step_many(acts)
    assert torch.equal(r.view(torch.int64), reward[T - 1].view(torch.int64)) and torch.equal(tm.to(torch.uint8), term[T - 1])
    c.check()
    got = _obs_np(c)
    for k in OBS_KEYS:
        assert np.array_equal(got[k], ob.tensors[k][T - 1].cpu().numpy()), f"step_many, no buffers: {k}"
    d2 = make(obs_layout=layout)
    d2.rollout(T, policy=2, policy_seed=99)
    d2.check()
    got = _obs_np(d2)
    for k in OBS_KEYS:
        assert np.array_equal(got[k], ob.tensors[k][T - 1].cpu().numpy()), f"rollout, no buffers: {k}"
    a.close(); b.close(); c.close(); d2.close()


@pytest.mark.parametrize("policy,scorer", [(2, False), (0, False), (2, True)])
def test_fused_rollout_vs_oracle(policy, scorer):
    """bg_rollout (policy on device, SAME_STEP auto-reset, every step's observation kept) against the oracle."""
    import torch
    from balatro_gym_amd.vec_env import ObsBuffers
    from oracle.gen_golden import IMPLEMENTED
    n, T = 256, 96
    seeds = [55_000 + SEED_OFFSET + 3 * i for i in range(n)]
    jokers = [random.Random(1000 + i).sample(IMPLEMENTED, 5) for i in range(n)] if scorer else None
    max_ante = 4 if scorer else 0
    env = _vec(n, seeds, scorer_jokers=scorer, autoreset=True, max_ante=max_ante)
    if jokers:
        env.inject(jokers=jokers, apply_now=True)
    ob = ObsBuffers(n, env.device, steps=T)
    reward = torch.zeros((T, n), dtype=torch.float64, device=env.device)
    term = torch.zeros((T, n), dtype=torch.uint8, device=env.device)
    acts = torch.zeros((T, n), dtype=torch.int32, device=env.device)
    env.rollout(T, policy=policy, policy_seed=4242, env_index0=17, t0=5, obs_buffers=ob, reward=reward, terminated=term,
                actions=acts)
    env.check()
    got_stats = env.stats()
    wobs, wr, wt, wa, wstats = _oracle_rollout(n, seeds, T, policy, 4242, scorer, max_ante, jokers, env_index0=17, t0=5)
    assert np.array_equal(acts.cpu().numpy(), wa)
    assert np.array_equal(term.cpu().numpy(), wt)
    assert np.array_equal(reward.cpu().numpy().view(np.uint64), wr.view(np.uint64))
    for k in OBS_KEYS:
        g = ob.tensors[k].cpu().numpy()
        assert np.array_equal(g, wobs[k]), f"rollout obs[{k}] differs at {np.argwhere(g != wobs[k])[0]}"
    for k in ("steps", "episodes", "plays", "score_sum", "reward_bits"):
        assert got_stats[k] == wstats[k], (k, got_stats[k], wstats[k])
    env.close()


@pytest.mark.parametrize("rings,async_refill", [("8,13,12", "1"), ("4,5,4", "1"), ("8,13,12", "0")])
def test_fused_rollout_shallow_rings(monkeypatch, rings, async_refill):
    """Same comparison with shallow look-ahead rings: one bg_rollout call becomes many launches with a refill between
    them (overlapped on the side streams, or synchronous), which must not change a single bit."""
    kg, ks, kd = rings.split(",")
    monkeypatch.setenv("BG_KG", kg); monkeypatch.setenv("BG_KS", ks); monkeypatch.setenv("BG_KD", kd)
    monkeypatch.setenv("BG_ASYNC_REFILL", async_refill)
    test_fused_rollout_vs_oracle(2, True)


def test_repeated_jokers_rollout_vs_oracle():
    """Envs that own a joker id more than once (what Ankh's copy leaves in `state.jokers`): two or three Bloodstones / 8 Balls /
    Triboulets per env.  Every copy draws per played card, so the chain's RNG offsets depend on all of them."""
    from balatro_gym_amd.vec_env import RowBuffers
    n, T = 256, 160
    seeds = [61_000 + SEED_OFFSET + 13 * i for i in range(n)]
    pool = [117, 117, 26, 26, 147, 116, 31, 1]
    jokers = [[random.Random(8000 + i).choice(pool) for _ in range(2 + i % 4)] for i in range(n)]
    env = _vec(n, seeds, scorer_jokers=True, autoreset=True, max_ante=4)
    env.inject(jokers=jokers, apply_now=True)
    rb = RowBuffers(n, env.device, steps=T)
    env.rollout(T, policy=2, policy_seed=77, obs_buffers=rb)
    env.check()
    got_stats = env.stats()
    wobs, wr, wt, wa, wstats = _oracle_rollout(n, seeds, T, 2, 77, True, 4, jokers)
    assert np.array_equal(rb.action.cpu().numpy(), wa)
    assert np.array_equal(rb.terminated.cpu().numpy(), wt)
    assert np.array_equal(rb.reward.contiguous().cpu().numpy().view(np.uint64), wr.view(np.uint64))
    for k in OBS_KEYS:
        assert np.array_equal(rb.tensors[k].contiguous().cpu().numpy(), wobs[k]), f"record key {k} differs"
    for k in ("steps", "episodes", "plays", "score_sum", "reward_bits"):
        assert got_stats[k] == wstats[k], (k, got_stats[k], wstats[k])
    assert got_stats["plays"] > 2000
    env.close()


def test_global_stream_budget_hungriest_policy(monkeypatch):
    """The block ring of the per-env global stream is priced at 32 words per step (bg_lib.hip `bg_gwords`; rounds 1-2: 110).  The policy
    that draws the most per step: five copies of a joker that draws per scoring card (8 Ball twice on an 8, Bloodstone), a play as soon as
    c cards are toggled (c = 1..5 by env), smallest blind, straight out of the shop.  Run through bg_step_many with the SHALLOWEST ring
    (BG_KG=5: four blocks ahead = a refill every 74 steps): no underflow (`check()`), bit-exact with the oracle, and the words every env
    really consumed (from its state blob) stay under the bound."""
    import torch
    monkeypatch.setenv("BG_KG", "5")
    n, T = 160, 296
    seeds = [77_000 + SEED_OFFSET + 3 * i for i in range(n)]
    jokers = [[26] * 5 if i % 2 else [117, 26, 117, 26, 27] for i in range(n)]

    def hungry(o, i, t):
        ob = o.obs()
        m, sel = ob["action_mask"], ob["selected_cards"]
        if int(ob["phase"]) == 0:
            if m[0] and int(np.count_nonzero(sel)) >= 1 + i % 5:
                return 0
            for k in range(8):
                kk = (k + i + t) % 8
                if m[2 + kk] and not sel[kk]:
                    return 2 + kk
            return 0 if m[0] else int(np.flatnonzero(m)[0])
        for a in (45, 31):
            if m[a]:
                return a
        return int(np.flatnonzero(m)[0])

    from balatro_gym_amd.vec_env import ObsBuffers
    wobs, wr, wt, wa, wstats = _oracle_rollout(n, seeds, T, 0, 0, True, 4, jokers, action_fn=hungry)
    assert wstats["plays"] > 4000
    env = _vec(n, seeds, scorer_jokers=True, autoreset=True, max_ante=4)
    env.inject(jokers=jokers, apply_now=True)
    ob = ObsBuffers(n, env.device, steps=T)
    reward = torch.zeros((T, n), dtype=torch.float64, device=env.device)
    term = torch.zeros((T, n), dtype=torch.uint8, device=env.device)
    env.step_many(torch.from_numpy(wa).to(env.device), obs_buffers=ob, reward=reward, terminated=term)
    env.check()
    assert np.array_equal(reward.cpu().numpy().view(np.uint64), wr.view(np.uint64))
    assert np.array_equal(term.cpu().numpy(), wt)
    for k in OBS_KEYS:
        assert np.array_equal(ob.tensors[k].cpu().numpy(), wobs[k]), f"obs[{k}] differs"
    words = np.array([env.parse_state_blob(env.get_state(i))["global_words_consumed"] for i in range(n)])
    print(f"global words per step: mean {words.mean() / T:.1f} max {words.max() / T:.1f}")
    assert words.max() <= 32 * T + 128, words.max()
    assert words.max() >= 8 * T, words.max()      # the policy really is hungry (a uniform policy draws 2-4 words per step)
    env.close()


@pytest.mark.parametrize("rings,async_refill", [("default", "1"), ("8,13,12", "1"), ("8,13,12", "0")])
def test_many_short_launches_vs_oracle(monkeypatch, rings, async_refill):
    """The same 120 steps as MANY short launches (7, 1, 20, 13, ... fused steps per bg_rollout_rows call): the look-ahead rings are
    topped up lazily -- only when the steps launched since the last refill could exhaust them, by a refill that runs beside the
    launch that needs it -- so most launches run without one.  Every record byte against the oracle, with deep and shallow rings."""
    from balatro_gym_amd.vec_env import RowBuffers
    from oracle.gen_golden import IMPLEMENTED
    if rings != "default":
        kg, ks, kd = rings.split(",")
        monkeypatch.setenv("BG_KG", kg); monkeypatch.setenv("BG_KS", ks); monkeypatch.setenv("BG_KD", kd)
    monkeypatch.setenv("BG_ASYNC_REFILL", async_refill)
    n, T = 300, 120
    seeds = [91_000 + SEED_OFFSET + 7 * i for i in range(n)]
    jokers = [random.Random(3000 + i).sample(IMPLEMENTED, 5) for i in range(n)]
    env = _vec(n, seeds, scorer_jokers=True, autoreset=True, max_ante=4)
    env.inject(jokers=jokers, apply_now=True)
    rb = RowBuffers(n, env.device, steps=T)
    sizes, done, k = [7, 1, 20, 13, 2, 30, 5, 17, 25], 0, 0
    while done < T:
        c = min(sizes[k % len(sizes)], T - done)
        part = RowBuffers.__new__(RowBuffers)  # a window of c rows of the same byte tensor
        part.n, part.steps, part.rows = n, c, rb.rows[done:done + c]
        env.rollout(c, policy=2, policy_seed=31, env_index0=9, t0=done, obs_buffers=part, zero_stats=(done == 0))
        done += c; k += 1
    env.check()
    got_stats = env.stats()
    wobs, wr, wt, wa, wstats = _oracle_rollout(n, seeds, T, 2, 31, True, 4, jokers, env_index0=9, t0=0)
    assert np.array_equal(rb.action.cpu().numpy(), wa)
    assert np.array_equal(rb.terminated.cpu().numpy(), wt)
    assert np.array_equal(rb.reward.contiguous().cpu().numpy().view(np.uint64), wr.view(np.uint64))
    for k in OBS_KEYS:
        g = rb.tensors[k].contiguous().cpu().numpy()
        assert np.array_equal(g, wobs[k]), f"record key {k} differs"
    for k in ("steps", "episodes", "plays", "score_sum", "reward_bits"):
        assert got_stats[k] == wstats[k], (k, got_stats[k], wstats[k])
    env.close()


@pytest.mark.parametrize("sliced,parts,sizes", [("1", None, (20, 13, 30, 7, 20, 20)), ("1", "16,16,16,16", (20, 13, 30, 7, 20, 20)), ("1", "1,1,1,1", (20, 13, 30, 7, 20, 20)),
                                                ("0", None, (20, 13, 30, 7, 20, 20)), ("1", None, (100, 180, 60, 20, 186, 150)), ("1", None, (180,))])
def test_sliced_refill_vs_oracle(monkeypatch, sliced, parts, sizes):
    """A LONG run of short launches with the default rings: 1 400 steps as 20 / 13 / 30 / 7-step bg_rollout_rows calls cross the point where the rings
    demand a refill several times.  Each of those refills is issued in PIECES (BG_REFILL_SLICED, the default): its scan beside the launch that demanded it,
    then one dense kernel over a part of a work list beside each of the next launches -- with the default parts, with 16 parts per list (a piece beside
    nearly every launch), with one part per list, with the pieces switched off, and with launches of up to half a refill period (100 / 180 / 186 steps:
    several pieces beside one launch, the first ones beside the launch that asked).  Every record byte and the statistics against the oracle; a step
    (bg_step: a synchronous refill has to issue what is left of the pieces first) and a state blob in the middle."""
    from balatro_gym_amd.vec_env import RowBuffers
    from oracle.gen_golden import IMPLEMENTED
    monkeypatch.setenv("BG_REFILL_SLICED", sliced)
    if parts:
        monkeypatch.setenv("BG_REFILL_PARTS", parts)
    n, T = 192, 1400
    seeds = [77_000 + SEED_OFFSET + 5 * i for i in range(n)]
    jokers = [random.Random(4100 + i).sample(IMPLEMENTED, 5) for i in range(n)]
    env = _vec(n, seeds, scorer_jokers=True, autoreset=True, max_ante=4)
    env.inject(jokers=jokers, apply_now=True)
    rb = RowBuffers(n, env.device, steps=T)
    done, k = 0, 0
    while done < T:
        c = min(sizes[k % len(sizes)], T - done)
        part = RowBuffers.__new__(RowBuffers)
        part.n, part.steps, part.rows = n, c, rb.rows[done:done + c]
        env.rollout(c, policy=2, policy_seed=57, env_index0=3, t0=done, obs_buffers=part, zero_stats=(done == 0))
        done += c; k += 1
        if k == 37 or (k == 5 and len(sizes) < 6):   # in the middle of a refill period: a state blob read (synchronises) -- the run must not notice
            env.get_state(5)
    env.check()
    got_stats = env.stats()
    wobs, wr, wt, wa, wstats = _oracle_rollout(n, seeds, T, 2, 57, True, 4, jokers, env_index0=3, t0=0)
    assert np.array_equal(rb.action.cpu().numpy(), wa)
    assert np.array_equal(rb.terminated.cpu().numpy(), wt)
    assert np.array_equal(rb.reward.contiguous().cpu().numpy().view(np.uint64), wr.view(np.uint64))
    for key in OBS_KEYS:
        assert np.array_equal(rb.tensors[key].contiguous().cpu().numpy(), wobs[key]), f"record key {key} differs"
    for key in ("steps", "episodes", "plays", "score_sum", "reward_bits"):
        assert got_stats[key] == wstats[key], (key, got_stats[key], wstats[key])
    env.close()


def test_packed_records_padded_stride():
    """bg_rollout_rows with padded records: the same 352 bytes per record as the dense layout.  Stride 384 is the FAST layout (every
    record written as three whole 128-byte lines): its bytes 352..383 are zeros; with any other stride (416 here) the bytes behind a
    record are left untouched."""
    import torch
    from balatro_gym_amd.vec_env import RowBuffers
    n, T = 300, 40
    seeds = [5_000 + SEED_OFFSET + i for i in range(n)]
    outs = []
    for stride in (0, 384, 416):
        env = _vec(n, seeds, scorer_jokers=True, autoreset=True, max_ante=4)
        rb = RowBuffers(n, env.device, steps=T, row_stride=stride)
        rb.rows.fill_(0xAB)
        env.rollout(T, policy=2, policy_seed=5, obs_buffers=rb)
        env.check()
        outs.append(rb.rows.cpu().numpy())
        env.close()
    assert outs[1].shape[-1] == 384 and np.array_equal(outs[0], outs[1][:, :, :352])
    assert (outs[1][:, :, 352:] == 0).all()
    assert outs[2].shape[-1] == 416 and np.array_equal(outs[0], outs[2][:, :, :352])
    assert (outs[2][:, :, 352:] == 0xAB).all()


@pytest.mark.parametrize("policy,scorer,n", [(0, False, 256), (2, True, 256), (2, True, 200), (0, True, 77)])
def test_packed_record_rollout_vs_oracle(policy, scorer, n):
    """bg_rollout_rows: one 352-byte record per (step, env); every key, the reward, the action and the terminated flag
    read back through the strided views must equal the oracle's, bit for bit (env counts that do not fill the last
    128-env workgroup included)."""
    from balatro_gym_amd.vec_env import RowBuffers
    from oracle.gen_golden import IMPLEMENTED
    T = 96
    seeds = [77_000 + SEED_OFFSET + 5 * i for i in range(n)]
    jokers = [random.Random(2000 + i).sample(IMPLEMENTED, 5) for i in range(n)] if scorer else None
    max_ante = 4 if scorer else 0
    env = _vec(n, seeds, scorer_jokers=scorer, autoreset=True, max_ante=max_ante)
    if jokers:
        env.inject(jokers=jokers, apply_now=True)
    rb = RowBuffers(n, env.device, steps=T)
    env.rollout(T, policy=policy, policy_seed=99, env_index0=3, t0=11, obs_buffers=rb)
    env.check()
    got_stats = env.stats()
    wobs, wr, wt, wa, wstats = _oracle_rollout(n, seeds, T, policy, 99, scorer, max_ante, jokers, env_index0=3, t0=11)
    assert np.array_equal(rb.action.cpu().numpy(), wa)
    assert np.array_equal(rb.terminated.cpu().numpy(), wt)
    assert np.array_equal(rb.reward.contiguous().cpu().numpy().view(np.uint64), wr.view(np.uint64))
    for k in OBS_KEYS:
        g = rb.tensors[k].contiguous().cpu().numpy()
        assert g.dtype == wobs[k].dtype and np.array_equal(g, wobs[k]), f"record key {k} differs"
    pad = rb.rows[:, :, 343:].cpu().numpy()
    assert not pad.any()
    for k in ("steps", "episodes", "plays", "score_sum", "reward_bits"):
        assert got_stats[k] == wstats[k], (k, got_stats[k], wstats[k])
    env.close()


@pytest.mark.parametrize("engine,cfg", [(1, None), (3, "413"), (3, "113"), (3, "113/64"), (3, "113/24"), (3, "213"), (3, "414")])
@pytest.mark.parametrize("stride", [352, 384])
def test_packed_record_rollout_every_engine(engine, cfg, stride, monkeypatch):
    """The step engines behind bg_rollout_rows -- bg_engine.h (workers + copiers) and bg_engine3.h (owner waves + service waves in one
    workgroup, in its workgroup shapes of 64 / 128 / 256 envs and with four service waves) -- against the oracle: every record byte of a fused
    rollout over several launches.  The handle reads BG_ENGINE / BG_E3_CFG when it is created (bg_create_ex), so each parameter really runs
    its shape; n = 333 leaves every shape a partial last workgroup (333 = 5 x 64 + 13 = 2 x 128 + 77 = 256 + 77)."""
    from balatro_gym_amd.vec_env import RowBuffers
    from oracle.gen_golden import IMPLEMENTED
    monkeypatch.setenv("BG_ENGINE", str(engine))
    if cfg:   # "113/k": k live envs per 64-env workgroup (BG_E3_EPW; by itself the library picks 8 for a job this small)
        monkeypatch.setenv("BG_E3_CFG", cfg.split("/")[0])
        if "/" in cfg:
            monkeypatch.setenv("BG_E3_EPW", cfg.split("/")[1])
    n, T, chunks = 333, 64, 3
    seeds = [88_000 + SEED_OFFSET + 3 * i for i in range(n)]
    jokers = [random.Random(2600 + i).sample(IMPLEMENTED, 5) for i in range(n)]
    env = _vec(n, seeds, scorer_jokers=True, autoreset=True, max_ante=4)
    env.inject(jokers=jokers, apply_now=True)
    rb = RowBuffers(n, env.device, steps=T * chunks, row_stride=stride)
    rbc = [RowBuffers(n, env.device, steps=T, row_stride=stride) for _ in range(chunks)]
    import torch
    gbuf = torch.zeros((1, n, 352), dtype=torch.uint8, device=env.device)
    if engine == 3:   # a one-rank "sharded job": the copy-out also writes every env's current record (each call's last step) into the gather buffer
        env.set_gather_peers([gbuf], 0)
    for c in range(chunks):   # several launches: hand-overs, images and queue positions carry over from one to the next
        env.rollout(T, policy=0, policy_seed=5, env_index0=1, t0=c * T, obs_buffers=rbc[c], zero_stats=(c == 0))
        if engine == 3:
            assert torch.equal(gbuf[0], rbc[c].rows[T - 1][:, :352]), f"launch {c}: gathered current records differ from the launch's last row"
    env.check()
    got_stats = env.stats()
    wobs, wr, wt, wa, wstats = _oracle_rollout(n, seeds, T * chunks, 0, 5, True, 4, jokers, env_index0=1, t0=0)
    for c in range(chunks):
        sl = slice(c * T, (c + 1) * T)
        assert np.array_equal(rbc[c].action.cpu().numpy(), wa[sl])
        assert np.array_equal(rbc[c].terminated.cpu().numpy(), wt[sl])
        assert np.array_equal(rbc[c].reward.contiguous().cpu().numpy().view(np.uint64), wr[sl].view(np.uint64))
        for k in OBS_KEYS:
            assert np.array_equal(rbc[c].tensors[k].contiguous().cpu().numpy(), wobs[k][sl]), f"launch {c}: record key {k} differs"
    for k in ("steps", "episodes", "plays", "score_sum", "reward_bits"):
        assert got_stats[k] == wstats[k], (k, got_stats[k], wstats[k])
    env.close()
    del rb


def test_engine_env_vars_are_per_handle_and_validated(monkeypatch):
    """BG_ENGINE / BG_E3_CFG are read by bg_create_ex, once per handle (they used to latch process-wide at the first launch, so a test
    that set them later silently ran the default shape), and an unknown value is refused instead of falling through to a default."""
    from balatro_gym_amd import _native as nat
    monkeypatch.setenv("BG_E3_CFG", "112")
    with pytest.raises(nat.NativeError, match="BG_E3_CFG"):
        _vec(64, [1 + i for i in range(64)])
    monkeypatch.delenv("BG_E3_CFG")
    monkeypatch.setenv("BG_ENGINE", "2")
    with pytest.raises(nat.NativeError, match="BG_ENGINE"):
        _vec(64, [1 + i for i in range(64)])


@pytest.mark.parametrize("cfg,n", [("213", 333), ("413", 333), ("414", 300), (None, 16385 + 63)])
def test_card_states_every_workgroup_shape_vs_oracle(cfg, n, monkeypatch):
    """The CARDS instantiation of bg_engine3_kernel in the shapes the small tests never reached (they all ran 64 envs per workgroup): 128 and
    256 envs per workgroup with a partial last workgroup, and the shape bg_lib.hip picks by itself above 16 384 envs (128 per workgroup) -- card states on
    half of every deck, two consumables per env and episode, every record byte of two launches against the oracle."""
    from balatro_gym_amd.vec_env import RowBuffers
    if cfg:
        monkeypatch.setenv("BG_E3_CFG", cfg)
    big = n > 4096
    T, chunks = (6, 2) if big else (48, 2)
    seeds = [93_000 + SEED_OFFSET + 5 * i for i in range(n)]
    jokers = [random.Random(5200 + i).sample(range(1, 151), 5) for i in range(n)]
    cons_ids = list(range(1, 23)) + list(range(30, 42)) + list(range(50, 68))
    cons = [[cons_ids[(3 * i) % 52], cons_ids[(11 * i + 5) % 52]] for i in range(n)]
    cards = []
    for i in range(n):
        rr = random.Random(6000 + i)
        cards.append([(d, rr.choice([0, 1, 4, 5, 6, 7, 8]), rr.choice([0, 0, 1]), rr.choice([0, 0, 1, 2, 3, 4])) for d in rr.sample(range(52), 26)])
    env = _vec(n, seeds, scorer_jokers=True, autoreset=True, max_ante=6, card_states=True)
    env.inject(jokers=jokers, apply_now=True)
    env.inject_cards(cards, apply_now=True)
    env.inject_consumables(cons, apply_now=True)
    rbc = [RowBuffers(n, env.device, steps=T, row_stride=384) for _ in range(chunks)]
    for c in range(chunks):
        env.rollout(T, policy=0, policy_seed=13, env_index0=2, t0=c * T, obs_buffers=rbc[c], zero_stats=(c == 0))
    env.check()
    got_stats = env.stats()
    wobs, wr, wt, wa, wstats = _oracle_rollout(n, seeds, T * chunks, 0, 13, True, 6, jokers, env_index0=2, cards=cards, consumables=cons)
    for c in range(chunks):
        sl = slice(c * T, (c + 1) * T)
        assert np.array_equal(rbc[c].action.cpu().numpy(), wa[sl])
        assert np.array_equal(rbc[c].reward.contiguous().cpu().numpy().view(np.uint64), wr[sl].view(np.uint64))
        for k in OBS_KEYS:
            assert np.array_equal(rbc[c].tensors[k].contiguous().cpu().numpy(), wobs[k][sl]), f"launch {c}: record key {k} differs"
    for k in ("steps", "episodes", "plays", "score_sum", "reward_bits"):
        assert got_stats[k] == wstats[k], (k, got_stats[k], wstats[k])
    env.close()


def test_configs1_all_4096_envs_vs_oracle():
    """BASELINE configs[1] at its full size and EVERY env against the oracle: 4 096 envs, vanilla deck, no jokers, Ante-1 three blinds (the blind by env
    index), two launches (48 + 20 steps) of packed 384-byte records -- the shape the library picks for a job this small (256 workgroups of 16 live envs)."""
    from balatro_gym_amd.vec_env import RowBuffers
    n, chunks = 4096, (48, 20)
    seeds = [1000 + SEED_OFFSET + i for i in range(n)]
    env = _vec(n, seeds, autoreset=True)
    rbc = [RowBuffers(n, env.device, steps=T, row_stride=384) for T in chunks]
    t0 = 0
    for T, rb in zip(chunks, rbc):
        env.rollout(T, policy=2, policy_seed=17, env_index0=0, t0=t0, obs_buffers=rb, zero_stats=(t0 == 0))
        t0 += T
    env.check()
    got_stats = env.stats()
    wobs, wr, wt, wa, wstats = _oracle_rollout(n, seeds, sum(chunks), 2, 17, False, 0, None)
    t0 = 0
    for T, rb in zip(chunks, rbc):
        sl = slice(t0, t0 + T)
        assert np.array_equal(rb.action.cpu().numpy(), wa[sl])
        assert np.array_equal(rb.terminated.cpu().numpy(), wt[sl])
        assert np.array_equal(rb.reward.contiguous().cpu().numpy().view(np.uint64), wr[sl].view(np.uint64))
        for k in OBS_KEYS:
            assert np.array_equal(rb.tensors[k].contiguous().cpu().numpy(), wobs[k][sl]), f"steps {t0}..{t0 + T}: record key {k} differs"
        t0 += T
    for k in ("steps", "episodes", "plays", "score_sum", "reward_bits"):
        assert got_stats[k] == wstats[k], (k, got_stats[k], wstats[k])
    env.close()


@pytest.mark.parametrize("config", ["configs2_jokers_antes_1_4", "configs3_share_consumables_all_jokers_antes_1_8", "configs4_full_game_curriculum"])
def test_full_size_slice_vs_oracle(config):
    """BASELINE.json's full size, compared with the ORACLE (not with itself): 65 536 envs on the benchmark's path -- 256 envs per workgroup,
    packed records 384 bytes apart, three launches of 20 fused steps (the driver's launch shape) -- and every record byte, reward bit
    pattern, action and terminated flag of 2 048 of them, spread over all 256 workgroups (eight per workgroup, a different lane and owner
    wave in each), against the oracle run on just those envs with their GLOBAL indexes.  configs[2]: 5 implemented jokers, Antes 1-4,
    blind by env index; configs[3]'s single-GPU share: 5 of all 150 joker ids, card states, two consumables per env and episode, Antes
    1-8, uniform policy; configs[4]'s single-GPU share: the full game -- uniform policy over every valid action incl. the boss blind (47: 28 boss types,
    boss_blinds.py:301-532) and shop buys / rerolls / sells (shop.py:160-205), 5 of all 150 joker ids, a PER-ENV curriculum cap (3..5) that rises by 3
    between the first and the second launch."""
    import torch
    from balatro_gym_amd.vec_env import RowBuffers
    from oracle.gen_golden import IMPLEMENTED
    n, T, chunks = 65536, 20, 3
    cons3 = config.startswith("configs3")
    full = config.startswith("configs4")
    seeds = [1000 + SEED_OFFSET + i for i in range(n)]
    pool = list(range(1, 151)) if (cons3 or full) else IMPLEMENTED
    jokers = [random.Random(i).sample(pool, 5) for i in range(n)]
    cons_ids = list(range(1, 23)) + list(range(30, 42)) + list(range(50, 68))
    cons = [[cons_ids[i % 52], cons_ids[(7 * i + 3) % 52]] for i in range(n)] if cons3 else None
    cards = [[(d, [0, 1, 4, 6, 8][(i + d) % 5], [0, 1, 3][d % 3], [0, 1, 2, 3, 4][(i // 3 + d) % 5]) for d in range(16)] if i % 3 == 0 else []
             for i in range(n)] if cons3 else None
    policy, max_ante = (0, 8) if cons3 else ((0, 3) if full else (2, 4))
    caps = np.array([3 + (i % 3) for i in range(n)], np.int32)
    env = _vec(n, seeds, autoreset=True, scorer_jokers=True, max_ante=max_ante, card_states=cons3)
    if cons3:
        env.inject_cards(cards, apply_now=True)
    env.inject(jokers=jokers, apply_now=True)
    if cons3:
        env.inject_consumables(cons, apply_now=True)
    if full:
        env.set_max_ante(caps)
    pick = np.array([32 * j + (5 * j + j // 8) % 32 for j in range(2048)], dtype=np.int64)   # 8 per 256-env workgroup, lanes and waves vary
    assert len(set(pick.tolist())) == 2048 and pick.max() < n
    pick_t = torch.from_numpy(pick).to(env.device)
    rb = RowBuffers(n, env.device, steps=T, row_stride=384)
    got = []
    for c in range(chunks):
        if full and c == 1:
            env.set_max_ante(caps + 3)   # the curriculum: every env's cap rises between two launches
        env.rollout(T, policy=policy, policy_seed=21, env_index0=0, t0=c * T, obs_buffers=rb, zero_stats=(c == 0))
        got.append(rb.rows[:, pick_t, :].cpu().numpy().copy())   # [T, 2048, 384]
    env.check()
    st = env.stats()
    env.close()
    del rb
    assert st["steps"] == n * T * chunks and st["plays"] > 0 and st["episodes"] > 0
    rows = np.concatenate(got, axis=0)
    sub = RowBuffers(len(pick), torch.device("cpu"), steps=T * chunks, row_stride=384)   # the same record layout over the slice
    sub.rows.copy_(torch.from_numpy(rows))
    wobs, wr, wt, wa, _ = _oracle_rollout(len(pick), [seeds[i] for i in pick], T * chunks, policy, 21, True, max_ante, [jokers[i] for i in pick],
                                          cards=[cards[i] for i in pick] if cons3 else None, consumables=[cons[i] for i in pick] if cons3 else None,
                                          env_indexes=pick.tolist(), caps={0: caps[pick], T: caps[pick] + 3} if full else None)
    if full:   # the workload is the one meant: boss blinds were played and shops were used (a buy / reroll leaves the shop phase with other money than it entered)
        assert (sub.tensors["boss_blind_active"].numpy() != 0).any() and (wa == 47).any() and ((wa >= 20) & (wa <= 30)).any()
    assert np.array_equal(sub.action.numpy(), wa)
    assert np.array_equal(sub.terminated.numpy(), wt)
    assert np.array_equal(sub.reward.contiguous().numpy().view(np.uint64), wr.view(np.uint64))
    for k in OBS_KEYS:
        assert np.array_equal(sub.tensors[k].contiguous().numpy(), wobs[k]), f"record key {k} differs"
    assert not rows[:, :, 343:].any()   # padding and the two zero pieces of the whole-line layout


def test_engines_agree_full_size(monkeypatch):
    """65 536 envs, configs[2]: the checksums of a fused rollout (every observation row hashed on the device, reward bits, episodes, plays,
    scores) must be the same whichever engine (and, for bg_engine3.h, service-wave count) ran it."""
    from balatro_gym_amd.vec_env import RowBuffers
    from oracle.gen_golden import IMPLEMENTED
    n, T = 65536, 40
    seeds = [1000 + SEED_OFFSET + i for i in range(n)]
    jokers = [random.Random(i).sample(IMPLEMENTED, 5) for i in range(n)]
    res = []
    for engine, cfg in ((1, "413"), (3, "413"), (3, "414")):
        monkeypatch.setenv("BG_ENGINE", str(engine))
        monkeypatch.setenv("BG_E3_CFG", cfg)
        env = _vec(n, seeds, scorer_jokers=True, autoreset=True, max_ante=4, fused_steps=T)
        env.inject(jokers=jokers, apply_now=True)
        rb = RowBuffers(n, env.device, steps=T, row_stride=384)
        for c in range(3):
            env.rollout(T, policy=2 | 0x100, policy_seed=7, t0=c * T, obs_buffers=rb, zero_stats=(c == 0))
        env.check()
        res.append(env.stats())
        env.close()
        del rb
    assert res[0] == res[1] == res[2], res


def test_packed_records_same_content_full_size():
    """At N = 65 536 the observation checksum (hash of every row, BG_POLICY_HASH_OBS) must not depend on the output
    layout, and the packed records must agree with the per-key arrays of the same rollout on every key."""
    import torch
    from balatro_gym_amd.vec_env import ObsBuffers, RowBuffers
    n, T = 65536, 24
    seeds = [1000 + SEED_OFFSET + i for i in range(n)]
    out = []
    for packed in (False, True):
        env = _vec(n, seeds, autoreset=True, fused_steps=T)
        ob = (RowBuffers if packed else ObsBuffers)(n, env.device, steps=T)
        env.rollout(T, policy=2 | 0x100, policy_seed=7, obs_buffers=ob)
        env.check()
        out.append((env.stats(), ob))
        env.close()
    assert out[0][0] == out[1][0]
    for k in OBS_KEYS:
        assert torch.equal(out[0][1].tensors[k], out[1][1].tensors[k].contiguous()), k


def test_card_states_rollout_vs_oracle():
    """Card states (BONUS / GLASS / STEEL / STONE / GOLD / LUCKY, FOIL, GOLD / RED / BLUE seals) on random deck indexes of
    every env, through the fused rollout with packed records: bit-exact against the oracle, resets included."""
    from balatro_gym_amd.vec_env import RowBuffers
    from oracle.gen_golden import IMPLEMENTED
    n, T = 200, 128
    seeds = [91_000 + SEED_OFFSET + 7 * i for i in range(n)]
    jokers = [random.Random(3000 + i).sample(IMPLEMENTED, 5) for i in range(n)]
    cards = []
    for i in range(n):
        rr = random.Random(4000 + i)
        cards.append([(d, rr.choice([0, 1, 4, 5, 6, 7, 8]), rr.choice([0, 0, 1]), rr.choice([0, 0, 1, 2, 3]))
                      for d in rr.sample(range(52), 20)])
    env = _vec(n, seeds, scorer_jokers=True, autoreset=True, max_ante=4, card_states=True)
    env.inject(jokers=jokers, apply_now=True)
    env.inject_cards(cards, apply_now=True)
    rb = RowBuffers(n, env.device, steps=T)
    env.rollout(T, policy=0, policy_seed=31, obs_buffers=rb)
    env.check()
    got_stats = env.stats()
    wobs, wr, wt, wa, wstats = _oracle_rollout(n, seeds, T, 0, 31, True, 4, jokers, cards=cards)
    assert np.array_equal(rb.action.cpu().numpy(), wa)
    assert np.array_equal(rb.terminated.cpu().numpy(), wt)
    assert np.array_equal(rb.reward.contiguous().cpu().numpy().view(np.uint64), wr.view(np.uint64))
    for k in OBS_KEYS:
        assert np.array_equal(rb.tensors[k].contiguous().cpu().numpy(), wobs[k]), f"record key {k} differs"
    for k in ("steps", "episodes", "plays", "score_sum", "reward_bits"):
        assert got_stats[k] == wstats[k], (k, got_stats[k], wstats[k])
    env.close()


def _run_consumables_rollout(n, T, seeds, scorer, jokers, cons, cards, pseed, min_uses):
    from balatro_gym_amd.vec_env import RowBuffers
    env = _vec(n, seeds, scorer_jokers=scorer, autoreset=True, max_ante=6, card_states=True)
    env.inject(jokers=jokers, apply_now=True)
    env.inject_cards(cards, apply_now=True)
    env.inject_consumables(cons, apply_now=True)
    rb = RowBuffers(n, env.device, steps=T)
    env.rollout(T, policy=0, policy_seed=pseed, obs_buffers=rb)
    env.check()
    got_stats = env.stats()
    wobs, wr, wt, wa, wstats = _oracle_rollout(n, seeds, T, 0, pseed, scorer, 6, jokers, cards=cards, consumables=cons)
    assert np.array_equal(rb.action.cpu().numpy(), wa)
    assert np.array_equal(rb.terminated.cpu().numpy(), wt)
    assert np.array_equal(rb.reward.contiguous().cpu().numpy().view(np.uint64), wr.view(np.uint64))
    for k in OBS_KEYS:
        assert np.array_equal(rb.tensors[k].contiguous().cpu().numpy(), wobs[k]), f"record key {k} differs"
    for k in ("steps", "episodes", "plays", "score_sum", "reward_bits"):
        assert got_stats[k] == wstats[k], (k, got_stats[k], wstats[k])
    assert ((wa >= 10) & (wa <= 14)).sum() > min_uses  # the consumable path was really exercised
    env.close()
    return wobs


@pytest.mark.parametrize("scorer", [False, True])
def test_consumables_rollout_vs_oracle(scorer):
    """Tarot / spectral / planet consumables (every one of the 52 ids, two per episode) plus purple / blue seals through the
    fused rollout with packed records: every record byte against the oracle, resets included."""
    n, T = 256, 160
    seeds = [93_000 + SEED_OFFSET + 5 * i for i in range(n)]
    pool = list(range(1, 23)) + list(range(30, 42)) + list(range(50, 68))
    jokers = [random.Random(5000 + i).sample(list(range(1, 151)), i % 6) for i in range(n)]
    cons = [[pool[i % len(pool)], random.Random(6000 + i).choice(pool)][: 2 - (i % 9 == 0)] for i in range(n)]
    cards = []
    for i in range(n):
        rr = random.Random(7000 + i)
        cards.append([(d, rr.choice([0, 0, 4, 8]), 0, rr.choice([0, 3, 4, 4])) for d in rr.sample(range(52), 16)] if i % 2 else [])
    _run_consumables_rollout(n, T, seeds, scorer, jokers, cons, cards, 47, 500)


def test_immolate_cryptid_rollout_vs_oracle():
    """The two consumables that change the deck length, over and over (The Fool copies them, purple seals bring more Fools):
    Immolate removes five sampled cards from the live deck list -- every later deck index, the hand's included, names another
    card, The Pillar's marks move along -- and Cryptid appends copies that only deck_size and Blue Joker ever see."""
    n, T = 192, 200
    seeds = [95_000 + SEED_OFFSET + 7 * i for i in range(n)]
    jokers = [[53, 1, 16][: 1 + i % 3] for i in range(n)]  # Blue Joker: +2 chips per card in the deck
    cons = [[[59, 65], [65, 59], [59, 1], [65, 1], [59, 59], [1, 59]][i % 6] for i in range(n)]
    cards = [[(d, [0, 7, 4][d % 3], 0, [0, 4][d % 2]) for d in range(20)] if i % 3 == 0 else [] for i in range(n)]
    wobs = _run_consumables_rollout(n, T, seeds, True, jokers, cons, cards, 53, 300)
    sizes = wobs["deck_size"]
    assert sizes.min() <= 42 and sizes.max() >= 54, (sizes.min(), sizes.max())  # decks really shrank and grew


def test_deck_length_fences_vs_oracle():
    """Where the deck-length consumables stop being followed (BG_ERR_CONSUMABLE_DECK), through bg_step in lockstep with the oracle: Immolate down to 12
    real cards (eight uses of a full deck; the ninth is refused with the state untouched), with Cryptid's copies in the deck before (random.sample's pool
    method then draws over real cards AND copies), alternating the two, and Cryptid alone up to 126 cards (the 38th use would make np.int8(128)).  One
    consumable is injected per use; observation, reward and error code of every step are compared."""
    import torch
    programs = [[59] * 10, [65, 65] + [59] * 10, [59, 65] * 8 + [59] * 4, [65] * 39, [65, 59, 59] * 5 + [59] * 3, [59] * 4 + [65] * 6 + [59] * 6]
    n = 2 * len(programs)
    seeds = [97_000 + SEED_OFFSET + 11 * i for i in range(n)]
    env = _vec(n, seeds, scorer_jokers=False, autoreset=False, max_ante=20, card_states=True)
    orc = _oracle_envs(n, seeds, False, 20)
    jokers = [[53] if i % 2 else [] for i in range(n)]   # Blue Joker counts the deck (copies included)
    env.inject(jokers=jokers, apply_now=True)
    for o, js in zip(orc, jokers):
        o.set_jokers(js)

    def step_all(acts, ctx):
        res = [o.step(int(a)) for o, a in zip(orc, acts)]
        _, reward, term, _, info = env.step(torch.from_numpy(np.asarray(acts, np.int32)).to(env.device))
        assert np.array_equal(reward.cpu().numpy().view(np.uint64), np.array([r[1] for r in res]).view(np.uint64)), ctx
        assert np.array_equal(info["error"].cpu().numpy(), np.array([r[4].error for r in res], dtype=np.int32)), ctx
        assert not term.cpu().numpy().any() and not any(r[2] for r in res), ctx
        _assert_obs(_obs_np(env), {k: np.stack([r[0][k] for r in res]) for k in OBS_KEYS}, ctx)
        return res

    step_all([45] * n, "blind select")
    refused, sizes = 0, set()
    for it in range(max(len(p) for p in programs)):
        cons = [[programs[i // 2][it]] if it < len(programs[i // 2]) else [] for i in range(n)]
        env.inject_consumables(cons, apply_now=True)
        for o, c in zip(orc, cons):
            o.set_consumables(c)
        for act in (2, 3):
            step_all([act] * n, f"use {it} select {act}")
        res = step_all([10 if c else 59 for c in cons], f"use {it}")
        refused += sum(r[4].error == 12 for r in res)
        sizes.update(int(r[0]["deck_size"]) for r in res)
    assert refused >= 6 and min(sizes) <= 12 and max(sizes) == 126, (refused, min(sizes), max(sizes))
    env.check()
    env.close()


def test_inject_consumables_needs_card_states():
    """Tarot / spectral ids are refused on a handle without card states (planets are fine): no silent divergence."""
    from balatro_gym_amd._native import NativeError
    env = _vec(4, [1, 2, 3, 4])
    env.inject_consumables([[30], [41, 35], [], [38]], apply_now=True)
    assert env.obs["consumables"].cpu().numpy()[:, :2].tolist() == [[30, 0], [41, 35], [0, 0], [38, 0]]
    with pytest.raises(NativeError):
        env.inject_consumables([[1], [], [], []])
    with pytest.raises(NativeError):
        env.inject_consumables([[23], [], [], []])
    env.close()


def test_service_wave_masks(monkeypatch):
    """The step engine with other waves made service-capable (BG_ENG_SMASK: waves 0, 1, 2 and 5 instead of 3..6) and with three
    instead of four of them: which waves own the RNG windows must not change a bit."""
    for mask in ("39", "112"):
        monkeypatch.setenv("BG_ENG_SMASK", mask)
        test_fused_rollout_vs_oracle(2, True)
    test_consumables_rollout_vs_oracle(True)


@pytest.mark.parametrize("copiers", ["1", "3"])
def test_copier_wave_counts(monkeypatch, copiers):
    """Packed records with ONE and with THREE copier waves instead of two (BG_ENG_COPIERS; the copy queue is dealt out in blocks of 32
    positions, round robin): every record byte against the oracle, dense and whole-line layouts, many short launches."""
    monkeypatch.setenv("BG_ENG_COPIERS", copiers)
    test_packed_record_rollout_vs_oracle(2, True, 256)
    test_packed_record_rollout_vs_oracle(0, True, 77)
    test_packed_records_padded_stride()
    test_many_short_launches_vs_oracle(monkeypatch, "8,13,12", "1")


@pytest.mark.parametrize("config", ["configs2_jokers_antes_1_4", "configs3_consumables_all_jokers_antes_1_8", "configs4_full_game_curriculum"])
def test_rollout_properties_full_size(config):
    """Size-independent properties at BASELINE.json's N = 65 536 on the REAL workloads -- configs[2] (5 random implemented jokers
    per env, scorer-level joker chain, Antes 1-4 cap, blind 45/46/47 by env index), the single-GPU share of configs[3] (full joker
    pool: 5 of all 150 ids, card states on, two consumables per env and episode out of all 52 planet / tarot / spectral ids, enhanced /
    sealed cards in a third of the decks, Antes 1-8 cap, uniform policy) and the single-GPU share of configs[4] (full
    game: uniform policy incl. boss blinds and shop buys / rerolls / sells, all 150 joker ids, a per-env curriculum cap that
    rises 3 -> 8 between launches): determinism, chunking invariance (one T=48 call == 48 T=1 calls), sharding invariance (two
    half-size handles with env_index0 offsets == one full handle), by statistics + XOR checksums of every reward and every
    observation record."""
    import torch
    from oracle.gen_golden import IMPLEMENTED
    n, T = 65536, 48
    seeds = [1000 + SEED_OFFSET + i for i in range(n)]
    full = config.startswith("configs4")
    cons3 = config.startswith("configs3")
    policy = (0 if (full or cons3) else 2) | 0x100
    pool = list(range(1, 151)) if (full or cons3) else IMPLEMENTED
    jokers = [random.Random(i).sample(pool, 5) for i in range(n)]
    caps = np.array([3 + (i % 3) for i in range(n)], np.int32)
    cons_ids = list(range(1, 23)) + list(range(30, 42)) + list(range(50, 68))   # _get_consumable_ids (balatro_env_2.py:1545-1567)
    cons = [[cons_ids[i % 52], cons_ids[(7 * i + 3) % 52]] for i in range(n)] if cons3 else None
    cards = [[(d, [0, 1, 4, 6, 8][(i + d) % 5], [0, 1, 3][d % 3], [0, 1, 2, 3, 4][(i // 3 + d) % 5]) for d in range(16)] if i % 3 == 0 else []
             for i in range(n)] if cons3 else None

    def run(count, index0, chunks):
        env = _vec(count, seeds[index0:index0 + count], autoreset=True, scorer_jokers=True, max_ante=8 if cons3 else (3 if full else 4),
                   fused_steps=48, card_states=cons3)
        if cons3:
            env.inject_cards(cards[index0:index0 + count], apply_now=True)
        env.inject(jokers=jokers[index0:index0 + count], apply_now=True)
        if cons3:
            env.inject_consumables(cons[index0:index0 + count], apply_now=True)
        t0 = 0
        for c in chunks:
            if full and t0 == 24:  # the curriculum: caps rise mid-run (CurriculumBalatroEnv.current_max_ante, per env)
                env.set_max_ante(caps[index0:index0 + count] + 3)
            env.rollout(c, policy=policy, policy_seed=7, env_index0=index0, t0=t0, zero_stats=(t0 == 0))
            t0 += c
        st = env.stats()
        final = {k: v.clone() for k, v in env.obs.items()}
        env.close()
        return st, final

    a, fa = run(n, 0, [24, 24])
    b, fb = run(n, 0, [24, 24])
    assert a == b, "rollout is not deterministic"
    assert a["steps"] == n * T and a["episodes"] > 0 and a["plays"] > 0
    c, fc = run(n, 0, [1] * T)
    assert a == c, "chunking changed the result"
    for k in OBS_KEYS:
        assert torch.equal(fa[k], fc[k]), k
    h = n // 2
    s0, f0 = run(h, 0, [24, 24])
    s1, f1 = run(h, h, [24, 24])
    for k in ("steps", "episodes", "plays", "score_sum"):
        assert s0[k] + s1[k] == a[k], k
    assert s0["reward_bits"] ^ s1["reward_bits"] == a["reward_bits"]
    assert s0["obs_hash"] ^ s1["obs_hash"] == a["obs_hash"]
    for k in OBS_KEYS:
        assert torch.equal(torch.cat([f0[k], f1[k]]), fa[k]), k


def test_curriculum_caps_vs_oracle():
    """bg_set_max_ante (CurriculumBalatroEnv.current_max_ante, train_balatro_agent.py:126-152): per-env caps set and RAISED in
    the middle of a fused rollout, against oracle envs whose caps change at the same steps; a template ante above the cap is
    refused."""
    import torch
    from balatro_gym_amd._native import NativeError
    from balatro_gym_amd.vec_env import RowBuffers
    from oracle import pyoracle as po
    n, T = 192, 90
    seeds = [61_000 + SEED_OFFSET + 3 * i for i in range(n)]
    caps0 = [1 + i % 3 for i in range(n)]
    caps1 = [c + 2 for c in caps0]
    env = _vec(n, seeds, autoreset=True, max_ante=0)
    env.inject(levels=np.full((n, 12), 15, np.uint8), apply_now=True)  # level-15 hands beat every blind at once: antes rise fast
    env.observe()
    env.set_max_ante(caps0)
    orc = [po.OracleEnv(s, max_ante=c) for s, c in zip(seeds, caps0)]

    def lv(o):
        for ht in range(12):
            o.set_hand_level(ht, 15)
    for o in orc:
        lv(o)
    got_r, got_t, want_r, want_t = [], [], [], []
    limit_hits = 0
    for part, caps in ((0, caps0), (1, caps1)):
        if part:
            env.set_max_ante(caps)
            for o, c in zip(orc, caps):
                o.set_max_ante(c)
        rb = RowBuffers(n, env.device, steps=T)
        env.rollout(T, policy=1, policy_seed=21, t0=part * T, obs_buffers=rb)
        got_r.append(rb.reward.contiguous().cpu().numpy()); got_t.append(rb.terminated.cpu().numpy())
        wr = np.zeros((T, n)); wt = np.zeros((T, n), np.uint8)
        for t in range(T):
            for i, o in enumerate(orc):
                _, r, term, _, info = o.step(o.policy_action(1, 21, i, part * T + t))
                wr[t, i], wt[t, i] = r, term
                limit_hits += bool(info.flags & 256)
                if term:
                    o.reset(); lv(o)
        want_r.append(wr); want_t.append(wt)
        last = {k: rb.tensors[k][T - 1].contiguous().cpu().numpy() for k in OBS_KEYS}
        for k in OBS_KEYS:
            assert np.array_equal(last[k], np.stack([o.obs()[k] for o in orc])), (part, k)
    for part in range(2):
        assert np.array_equal(got_t[part], want_t[part]), part
        assert np.array_equal(got_r[part].view(np.uint64), want_r[part].view(np.uint64)), part
    assert limit_hits > 20, limit_hits  # episodes really ended at their cap
    env.set_max_ante(4)
    with pytest.raises(NativeError, match="cap"):
        env.inject(ante=[5] * n)
    # PER-ENV caps on a handle created without one (what CurriculumTracker does): template antes are validated against each env's own
    # cap, in both orders -- an env that resets above its cap would end an episode per step and drain the deck ring
    per_env = [2 + i % 3 for i in range(n)]
    env.set_max_ante(per_env)
    env.inject(ante=per_env)                                   # exactly at the cap: fine
    with pytest.raises(NativeError, match="cap"):
        env.inject(ante=[c + 1 for c in per_env])
    with pytest.raises(NativeError, match="template ante"):
        env.set_max_ante([c - 1 for c in per_env])
    env.rollout(12, policy=0, policy_seed=5)
    env.check()
    env.close()


def test_single_env_info_dict_vs_reference():
    """The info dict of the N = 1 drop-in (balatro_env_2.py:894-925, :1283-1288): the boss blinds' formatted rejection messages
    (boss_blinds.py:388-405), 'score_breakdown' / 'reward_breakdown' / 'final_score' / 'hand_type' / 'cards_played' of a play,
    'beat_blind' / 'failed', 'boss_blind' -- against what the reference returned (golden trace boss_forced: The Psychic, The Eye, The
    Mouth and The Verdant each reject at least once in the seeds replayed here)."""
    from balatro_gym_amd import BalatroEnv
    tr = load_trace("boss_forced")
    first = tr["obs_boss_blind_type"][:, 0].astype(int)
    seen = set()
    for boss in (7, 12, 13, 25, 2):
        for si in np.nonzero(first == boss)[0][:4]:
            env = BalatroEnv(seed=int(tr["seeds"][si]))
            for t in range(tr["actions"].shape[1]):
                obs, r, term, trunc, info = env.step(int(tr["actions"][si, t]))
                ctx = (boss, int(si), t)
                assert r == tr["rewards"][si, t] and term == bool(tr["terminated"][si, t]), ctx
                msg = str(tr["error_msg"][si, t])
                assert info.get("error", "") == msg, (ctx, info.get("error"), msg)
                if msg:
                    seen.add(msg.split(" ")[0] + " " + msg.split(" ")[1])
                assert info.get("boss_blind", "") == str(tr["boss_blind"][si, t]), ctx
                assert bool(info.get("beat_blind")) == bool(tr["beat_blind"][si, t]) and bool(info.get("failed")) == bool(tr["failed"][si, t]), ctx
                if tr["hand_type"][si, t] >= 0:
                    bi, bf = tr["breakdown_int"][si, t], tr["breakdown_f64"][si, t]
                    want = {"base_chips": int(bi[0]), "base_mult": int(bi[1]), "card_chips": int(bi[2]), "joker_chips": int(bi[3]),
                            "joker_mult": int(bi[4]), "joker_x_mult": float(bf[0]), "final_chips": int(bi[5]), "final_mult": int(bi[6]),
                            "final_x_mult": float(bf[1]), "final_score": int(int(bi[5]) * int(bi[6]) * float(bf[1])), "money_gained": int(bi[7])}
                    assert info["score_breakdown"] == want, (ctx, info["score_breakdown"], want)
                    assert info["final_score"] == tr["final_score"][si, t] and info["cards_played"] == tr["cards_played"][si, t], ctx
                    assert [info["reward_breakdown"][k] for k in ("progress", "milestone", "score", "hand_quality", "efficiency", "synergy",
                                                                   "strategy", "ante_bonus")] == tr["reward_terms"][si, t].tolist(), ctx
                else:
                    assert "score_breakdown" not in info, ctx
                if term:
                    env.reset()
            env.close()
    assert {"Must play", "Cannot play", "Can only"} <= seen, seen


def test_single_env_gym_surface():
    """BalatroEnv (N = 1) keeps the reference surface: config 1 of BASELINE.json against the golden trace."""
    from balatro_gym_amd import BalatroEnv
    tr = load_trace("c1_small_only")
    si = 0
    env = BalatroEnv(seed=int(tr["seeds"][si]))
    assert env.action_space.n == 60
    # the space DECLARES the reference's 51 keys (balatro_env_2.py:386-470); an observation holds the 31 `_get_observation` fills (Q14)
    assert list(env.observation_space.spaces.keys())[:31] == list(OBS_KEYS) and len(env.observation_space.spaces) == 51
    obs = env._np_obs()
    assert set(obs) == set(OBS_KEYS)
    for k in OBS_KEYS:
        assert np.array_equal(obs[k], tr["obs0_" + k][si]) and np.asarray(obs[k]).dtype == tr["obs0_" + k].dtype, k
    for t in range(200):
        obs, r, term, trunc, info = env.step(int(tr["actions"][si, t]))
        assert isinstance(r, float) and r == tr["rewards"][si, t]
        assert term == bool(tr["terminated"][si, t]) and trunc is False
        for k in OBS_KEYS:
            assert np.array_equal(obs[k], tr["obs_" + k][si, t]), (t, k)
        if "final_score" in info:
            assert info["final_score"] == tr["final_score"][si, t]
        if term:
            obs, inf = env.reset()
            assert inf == {}
    # reset(seed=s) reproduces shuffle #1
    obs, _ = env.reset(seed=int(tr["seeds"][si]))
    env.step(45)
    first = env._np_obs()["hand"].tolist()
    env2 = BalatroEnv(seed=int(tr["seeds"][si]))
    env2.step(45)
    assert env2._np_obs()["hand"].tolist() == first
    env.close(); env2.close()


@pytest.mark.parametrize("cards", [False, True])
def test_save_load_state_roundtrip(cards):
    import torch
    n = 64
    env = _vec(n, [300 + i for i in range(n)], autoreset=True, scorer_jokers=cards, card_states=cards)
    if cards:  # the blob then also carries the card states, their reset copy and the 'card_enhancement' stream
        env.inject(jokers=[[1, 61, 27][: 1 + i % 3] for i in range(n)], apply_now=True)
        env.inject_cards([[(d, [4, 8, 5, 1][d % 4], d % 2, d % 4) for d in range(0, 52, 3)] for _ in range(n)])
    env.rollout(40, policy=0, policy_seed=3)
    blob = env.get_state(5)
    env.rollout(25, policy=0, policy_seed=3, t0=40)
    after = {k: v[5].clone() for k, v in env.obs.items()}
    env.set_state(5, blob)
    env.rollout(25, policy=0, policy_seed=3, t0=40)
    for k in OBS_KEYS:
        assert torch.equal(env.obs[k][5], after[k]), k
    env.check()
    env.close()


def test_state_blob_into_another_handle_vs_oracle(monkeypatch):
    """save_state / load_state (balatro_env_2.py:1575-1615) as blobs that travel: the state of env 5 of one handle after 60
    steps is restored into env 2 of ANOTHER handle (other size, other seeds), and that env then continues in lockstep with the
    oracle env that took the same 60 steps -- every observation key, reward bits, terminated, resets included.  Blobs that do
    not fit (truncated, another version, other ring depths, a handle without card states) are refused with a message."""
    import torch
    from balatro_gym_amd._native import NativeError
    from oracle import pyoracle as po
    from oracle.gen_golden import IMPLEMENTED
    nA, src, K, M = 16, 5, 60, 120
    seedsA = [41_000 + 3 * i for i in range(nA)]
    jok = [random.Random(900 + i).sample(IMPLEMENTED, 5) for i in range(nA)]
    A = _vec(nA, seedsA, scorer_jokers=True, autoreset=False, max_ante=4)
    A.inject(jokers=jok, apply_now=True)
    orc = po.OracleEnv(seedsA[src], scorer_jokers=True, max_ante=4)
    orc.set_jokers(jok[src])
    for t in range(K):
        a_src = orc.policy_action(0, 13, src, t)
        acts = np.full(nA, 59, np.int32)  # the other envs only send an action that is never valid
        acts[src] = a_src
        _, _, term, _, _ = A.step(torch.from_numpy(acts).to(A.device))
        _, _, ot, _, _ = orc.step(a_src)
        assert bool(term[src].item()) == ot
        if ot:
            orc.reset(); orc.set_jokers(jok[src])
            m = np.zeros(nA, np.uint8); m[src] = 1
            A.reset(mask=torch.from_numpy(m).to(A.device))
    blob = A.get_state(src)
    A.close()
    nB, dst = 7, 2
    B = _vec(nB, [77 + i for i in range(nB)], scorer_jokers=True, autoreset=False, max_ante=4)
    B.set_state(dst, blob)
    B.observe()
    want = orc.obs()
    for k in OBS_KEYS:
        assert np.array_equal(B.obs[k][dst].cpu().numpy(), want[k]), k
    for t in range(K, K + M):
        a_dst = orc.policy_action(0, 13, src, t)
        acts = np.full(nB, 59, np.int32)
        acts[dst] = a_dst
        _, reward, term, _, info = B.step(torch.from_numpy(acts).to(B.device))
        ob, r, ot, _, oi = orc.step(a_dst)
        ctx = f"t {t} action {a_dst}"
        assert np.float64(reward[dst].item()).view(np.uint64) == np.float64(r).view(np.uint64), ctx
        assert bool(term[dst].item()) == ot and int(info["final_score"][dst].item()) == oi.final_score, ctx
        for k in OBS_KEYS:
            assert np.array_equal(B.obs[k][dst].cpu().numpy(), ob[k]), f"{ctx}: {k}"
        if ot:
            ob = orc.reset(); orc.set_jokers(jok[src])
            m = np.zeros(nB, np.uint8); m[dst] = 1
            B.reset(mask=torch.from_numpy(m).to(B.device))   # the reset template (jokers) travelled in the blob
            ob = orc.obs()
            for k in OBS_KEYS:
                assert np.array_equal(B.obs[k][dst].cpu().numpy(), ob[k]), f"{ctx} reset: {k}"
    B.check()
    # blobs that do not fit
    with pytest.raises(NativeError, match="bytes"):
        B.set_state(dst, blob[:-16])
    bad = bytearray(blob); bad[4] ^= 0x7f
    with pytest.raises(NativeError, match="version"):
        B.set_state(dst, bytes(bad))
    bad = bytearray(blob); bad[0] ^= 0xff
    with pytest.raises(NativeError, match="magic"):
        B.set_state(dst, bytes(bad))
    with pytest.raises(NativeError, match="env_index"):
        B.set_state(nB, blob)
    B.close()
    monkeypatch.setenv("BG_KG", "9"); monkeypatch.setenv("BG_KS", "13"); monkeypatch.setenv("BG_KD", "12")
    C_ = _vec(4, [1, 2, 3, 4], scorer_jokers=True, autoreset=False, max_ante=4)
    with pytest.raises(NativeError, match="bytes|ring depths"):
        C_.set_state(0, blob)
    small = C_.get_state(0)
    C_.close()
    monkeypatch.delenv("BG_KG"); monkeypatch.delenv("BG_KS"); monkeypatch.delenv("BG_KD")
    D_ = _vec(4, [1, 2, 3, 4], scorer_jokers=True, autoreset=False, max_ante=4)
    with pytest.raises(NativeError, match="bytes|ring depths"):
        D_.set_state(0, small + bytes(len(blob)))  # long enough, but saved with other ring depths
    D_.close()
    E_ = _vec(4, [1, 2, 3, 4], scorer_jokers=True, autoreset=False, max_ante=4, card_states=True)
    with pytest.raises(NativeError, match="bytes|card states"):
        E_.set_state(0, blob)
    E_.close()


def test_state_blob_mid_sliced_refill_vs_oracle(monkeypatch):
    """A state blob taken while a refill is still being issued IN PIECES (bg_refill_pieces: the scan has advanced the producer counters, the dense
    kernels that fill those ring slots are handed out over the next launches).  bg_get_state / bg_set_state must issue what is pending first: the blob
    would otherwise claim shop-stream slots that hold the streams of 12 shops ago, and pieces issued after a bg_set_state would overwrite the restored
    env's slots with the replaced env's seeds.  Shallow rings (a refill every third 5-step launch, every slot reached within ~150 steps); env j's blob is
    taken behind launch j + 1 -- every phase of the period -- restored into ANOTHER handle and continued for 200 steps (~14 shops) against the oracle."""
    from balatro_gym_amd.vec_env import RowBuffers
    from oracle.gen_golden import IMPLEMENTED
    monkeypatch.setenv("BG_KG", "8"); monkeypatch.setenv("BG_KS", "13"); monkeypatch.setenv("BG_KD", "12")
    monkeypatch.setenv("BG_REFILL_SLICED", "1")
    n, L, K, M = 64, 5, 24, 200
    seeds = [83_000 + SEED_OFFSET + 11 * i for i in range(n)]
    jokers = [random.Random(5200 + i).sample(IMPLEMENTED, 5) for i in range(n)]
    A = _vec(n, seeds, scorer_jokers=True, autoreset=True, max_ante=4)
    assert L <= A.max_fused_steps // 2, A.max_fused_steps   # short enough for the refill to be issued in pieces
    A.inject(jokers=jokers, apply_now=True)
    rbA = RowBuffers(n, A.device, steps=L)
    blobs = []
    for j in range(K):
        A.rollout(L, policy=2, policy_seed=57, env_index0=0, t0=L * j, obs_buffers=rbA, zero_stats=(j == 0))
        blobs.append(A.get_state(j))
    A.check()
    A.close()
    wobs, wr, wt, wa, _ = _oracle_rollout(K, seeds[:K], L * K + M, 2, 57, True, 4, jokers[:K])
    B = _vec(n, [5 + i for i in range(n)], scorer_jokers=True, autoreset=True, max_ante=4)
    rbB = RowBuffers(n, B.device, steps=M)
    for j in range(K):
        t0 = L * (j + 1)
        B.set_state(j, blobs[j])
        B.rollout(M, policy=2, policy_seed=57, env_index0=0, t0=t0, obs_buffers=rbB, zero_stats=True)
        B.check()
        ctx = f"blob of env {j} taken behind launch {j + 1}"
        assert np.array_equal(rbB.action[:, j].cpu().numpy(), wa[t0:t0 + M, j]), ctx
        assert np.array_equal(rbB.terminated[:, j].cpu().numpy(), wt[t0:t0 + M, j]), ctx
        assert np.array_equal(rbB.reward[:, j].contiguous().cpu().numpy().view(np.uint64), wr[t0:t0 + M, j].view(np.uint64)), ctx
        for key in OBS_KEYS:
            g, w = rbB.tensors[key][:, j].contiguous().cpu().numpy(), wobs[key][t0:t0 + M, j]
            assert np.array_equal(g, w), f"{ctx}: record key {key} differs at step {t0 + int(np.argwhere(g.reshape(M, -1) != w.reshape(M, -1))[0][0])}"
    B.close()


def test_invalid_actions_and_no_raise():
    """Invalid actions never raise: reward -1.0, state unchanged (balatro_env_2.py:626-627)."""
    import torch
    n = 128
    env = _vec(n, [77 + i for i in range(n)], autoreset=False)
    before = {k: v.clone() for k, v in env.obs.items()}
    for bad in (0, 1, 31, 59, -1, 60, 1000):
        _, r, term, _, info = env.step(torch.full((n,), bad, dtype=torch.int32, device=env.device))
        assert (r == -1.0).all() and not term.any() and (info["error"] == 1).all()
        for k in OBS_KEYS:
            assert torch.equal(env.obs[k], before[k]), (bad, k)
    env.close()


def test_sb3_adapter_vs_reference_wrappers():
    """BalatroSB3VecEnv against the reference's OWN wrappers: SafeBalatroEnv(BalatroEnvFixed(seed + rank)) stepped like an SB3
    VecEnv (tests/golden/sb3_fixed.npz, generated from train_balatro_fixed.py:20-288): all 51 keys of every fixed observation,
    float32 rewards (incl. the -50 of an invalid-action termination), dones, the wrapper's info flags, the terminal
    observation of every wrapper-made ending, and the first observation of the episode that follows a game over."""
    from balatro_gym_amd.sb3_adapter import BalatroSB3VecEnv
    from tests.helpers import GOLD
    import os
    with np.load(os.path.join(GOLD, "sb3_fixed.npz")) as z:
        g = {k: z[k] for k in z.files}
    keys = [str(k) for k in g["keys"]]
    S, T = g["actions"].shape
    venv = BalatroSB3VecEnv(S, seed=int(g["seed0"]), max_invalid_actions=int(g["max_invalid_actions"]),
                            max_episode_steps=int(g["max_episode_steps"]))
    obs = venv.reset()
    for k in keys:
        assert obs[k].dtype == g["obs0_" + k].dtype and np.array_equal(obs[k], g["obs0_" + k]), k
    wrapper_ends = game_overs = 0
    for t in range(T):
        obs, rew, done, infos = venv.step(g["actions"][:, t])
        ctx = f"t {t}"
        assert rew.dtype == np.float32 and np.array_equal(rew.view(np.uint32), g["rewards"][:, t].view(np.uint32)), ctx
        assert np.array_equal(done, g["dones"][:, t].astype(bool)), ctx
        for k in keys:
            assert np.array_equal(obs[k], g["obs_" + k][:, t]), f"{ctx}: obs[{k}]"
        for i in range(S):
            assert bool(infos[i].get("invalid_action_termination")) == bool(g["invalid_action_termination"][i, t]), (ctx, i)
            assert bool(infos[i].get("max_steps_reached")) == bool(g["max_steps_reached"][i, t]), (ctx, i)
            if done[i]:
                assert infos[i]["TimeLimit.truncated"] == bool(g["truncated"][i, t] and not g["invalid_action_termination"][i, t]), (ctx, i)
                if g["invalid_action_termination"][i, t] or (g["truncated"][i, t] and not g["terminated"][i, t]):
                    wrapper_ends += 1
                    tob = infos[i]["terminal_observation"]
                    for k in keys:
                        assert np.array_equal(tob[k], g["term_" + k][i, t]), f"{ctx} env {i}: terminal_observation[{k}]"
                else:
                    game_overs += 1
    assert wrapper_ends > 100 and game_overs > 10, (wrapper_ends, game_overs)
    venv.close()


def test_sb3_adapter_conventions():
    """BalatroSB3VecEnv: VecEnv calling convention, BalatroEnvFixed's 51-key observation, SafeBalatroEnv's endings
    (N consecutive invalid actions -> terminated with -50, step limit -> truncated + terminal_observation)."""
    import torch
    from balatro_gym_amd.sb3_adapter import FIXED_SPEC, BalatroSB3VecEnv
    n = 64
    venv = BalatroSB3VecEnv(n, seed=300, max_invalid_actions=5, max_episode_steps=40)
    obs = venv.reset()
    assert set(obs) == set(FIXED_SPEC)
    for k, (dt, shape, _) in FIXED_SPEC.items():
        assert obs[k].shape == (n,) + shape and obs[k].dtype == np.dtype(dt), k
    # the fixed observation is the raw one, reshaped / retyped
    raw = {k: v.cpu().numpy() for k, v in venv.env.obs.items()}
    for k in OBS_KEYS:
        assert np.array_equal(obs[k].reshape(n, -1).astype(np.int64), raw[k].reshape(n, -1).astype(np.int64)), k
    # env 0 only sends an invalid action (59); the others toggle card 0 for ever (first valid action elsewhere)
    kills = truncs = 0
    for t in range(45):
        mask = obs["action_mask"]
        act = np.array([2 if mask[i, 2] else int(np.flatnonzero(mask[i])[0]) for i in range(n)], dtype=np.int64)
        act[0] = 59
        obs, rew, done, infos = venv.step(act)
        assert rew.dtype == np.float32 and done.dtype == np.bool_ and len(infos) == n
        if infos[0].get("invalid_action_termination"):
            kills += 1
            assert rew[0] == -50.0 and done[0] and "terminal_observation" in infos[0] and not infos[0]["TimeLimit.truncated"]
            assert (t + 1) % 5 == 0
        for i in range(1, n):
            if infos[i].get("max_steps_reached"):
                truncs += 1
                assert done[i] and infos[i]["TimeLimit.truncated"] and set(infos[i]["terminal_observation"]) == set(FIXED_SPEC)
                assert obs["phase"][i, 0] == 2  # a fresh episode: blind select
    assert kills == 9 and truncs > 0
    venv.close()


def test_mt_streams_on_device():
    """F1 on the DEVICE (DeterministicRNG balatro_env_2.py:84-144): the reference's known answers of tests/golden/mt_streams.json
    read back from the HIP state through bg_get_state blobs -- the FULL 52-card deck of the first reset shuffle (stream 0; an observation
    only ever shows deck[0..7]), the first three `get_int('shop_generation', 0, 2**31 - 1)` draws (stream 2) as the seeds of the first
    pre-seeded shop streams, and the first 16 words of random.Random(seed) as the head of the env's global-stream ring (raw MT19937
    words, tempered here as genrand_uint32 does)."""
    import json
    from balatro_gym_amd import BalatroVecEnv
    from tests.helpers import GOLD
    g = json.load(open(os.path.join(GOLD, "mt_streams.json")))
    seeds = [int(s) for s in g["seeds"]]
    n = len(seeds)

    def temper(y):
        y = y.astype(np.uint32).copy()
        y ^= y >> 11
        y ^= (y << 7) & np.uint32(0x9D2C5680)
        y ^= (y << 15) & np.uint32(0xEFC60000)
        y ^= y >> 18
        return y

    env = _vec(n, seeds, autoreset=False)   # construction = DeterministicRNG(seed) + the first reset()
    for i, seed in enumerate(seeds):
        st = BalatroVecEnv.parse_state_blob(env.get_state(i))
        assert st["deck"].tolist() == g["deck"][i], f"seed {seed}: the shuffled deck differs"
        cur = st["shop_slot_current"]
        got = [int(st["shop_slot_seeds"][(cur + 1 + k) % st["KS"]]) for k in range(3)]
        assert got == g["shop_seed"][i], f"seed {seed}: shop seeds {got} vs {g['shop_seed'][i]}"
    env.close()
    # the global stream is random.Random((seed + 16000) % 2**32) (harness convention G): an env seeded s - 16000 owns random.Random(s)
    small = [(i, s) for i, s in enumerate(seeds) if 16000 <= s < 2 ** 32]
    env = _vec(len(small), [s - 16000 for _, s in small], autoreset=False)
    for k, (i, seed) in enumerate(small):
        st = BalatroVecEnv.parse_state_blob(env.get_state(k))
        hot6 = st["hot"][6]
        g_cur = int((hot6[3] >> 16) & 0xff)   # ring block the cursor is in (nothing has drawn from the global stream yet)
        got = temper(st["global_blocks"][g_cur][:16]).tolist()
        assert got == g["u32"][i], f"seed {seed}: global stream words"
    env.close()


def test_step_rows_layout_vs_oracle():
    """`BalatroVecEnv(obs_layout="rows")`: bg_step_rows / bg_observe_rows -- the observation as one packed 384-byte record per env, written by
    the copier waves, `obs[key]` strided views of it -- in lockstep with the oracle (every key, reward bits, terminated, the info arrays),
    masked resets and an injection included; and against the "keys" layout of the same env."""
    import torch
    from oracle import pyoracle as po
    from oracle.gen_golden import IMPLEMENTED
    n, T = 320, 160
    seeds = [733_000 + SEED_OFFSET + 5 * i for i in range(n)]
    jokers = [random.Random(i).sample(IMPLEMENTED, i % 6) for i in range(n)]
    env = _vec(n, seeds, scorer_jokers=True, autoreset=False, max_ante=4, obs_layout="rows")
    twin = _vec(n, seeds, scorer_jokers=True, autoreset=False, max_ante=4)
    for e in (env, twin):
        e.inject(jokers=jokers, apply_now=True)
        e.observe()
    orc = _oracle_envs(n, seeds, True, 4, jokers)
    assert env.obs_rows.shape == (n, 384) and (env.obs_rows[:, 352:] == 0).all()
    _assert_obs(_obs_np(env), {k: np.stack([o.obs()[k] for o in orc]) for k in OBS_KEYS}, "initial")
    for t in range(T):
        acts = np.array([o.policy_action(0, 99, i, t) for i, o in enumerate(orc)], dtype=np.int32)
        res = [o.step(int(a)) for o, a in zip(orc, acts)]
        a = torch.from_numpy(acts).to(env.device)
        _, reward, term, _, info = env.step(a)
        _, reward2, term2, _, info2 = twin.step(a)
        ctx = f"rows layout t {t}"
        wr = np.array([r[1] for r in res])
        assert np.array_equal(reward.cpu().numpy().view(np.uint64), wr.view(np.uint64)), ctx
        wt = np.array([r[2] for r in res], dtype=np.uint8)
        assert np.array_equal(term.cpu().numpy(), wt), ctx
        for k in ("final_score", "error", "hand_type", "cards_played", "aux"):
            assert torch.equal(info[k], info2[k]), (ctx, k)
        assert np.array_equal(info["error"].cpu().numpy(), np.array([r[4].error for r in res], dtype=np.int32)), ctx
        assert torch.equal(info["reward_terms"].view(torch.int64), info2["reward_terms"].view(torch.int64)), ctx
        _assert_obs(_obs_np(env), {k: np.stack([r[0][k] for r in res]) for k in OBS_KEYS}, ctx)
        rows = env.obs_rows.cpu().numpy()
        assert np.array_equal(rows[:, 136:144].copy().view(np.float64)[:, 0].view(np.uint64), wr.view(np.uint64)), ctx   # reward rides in the record
        assert np.array_equal(rows[:, 172:176].copy().view(np.int32)[:, 0], acts) and np.array_equal(rows[:, 342], wt), ctx
        if wt.any():
            for i in np.nonzero(wt)[0]:
                orc[i].reset()
                orc[i].set_jokers(jokers[i])
            m = torch.from_numpy(wt).to(env.device)
            env.reset(mask=m); twin.reset(mask=m)
            _assert_obs(_obs_np(env), {k: np.stack([o.obs()[k] for o in orc]) for k in OBS_KEYS}, ctx + " after reset")
    env.check(); twin.check()
    env.close(); twin.close()
