"""Harness that imports the Python reference (BUILD-CONTAINER ONLY; TEST INFRASTRUCTURE).

The reference (`/root/reference`, read-only) is pure Python and needs `gymnasium`, which this image lacks.
No arithmetic lives in gymnasium, so a ~25-line stand-in module is registered before the import (SURVEY.md
Appendix C).  Nothing of the reference is copied: it is imported from where it lies, and only input/output
VECTORS produced with it are committed (tests/golden/, see oracle/gen_golden.py).

Harness conventions the reference leaves open (also restated in DESIGN.md):
  * per-env "global random": the reference shares the process-wide `random` module between all envs; here
    every env owns a stream, seeded `random.seed(G(seed))`, G(seed) = (seed + 16000) mod 2**32, that is swapped
    in and out around every call into that env.
  * scorer-level joker semantics (flag `scorer_jokers`): `UnifiedGameState.to_dict()` hands the scorer joker
    NAMES (as unified_scoring.py:313-351 does) instead of dicts, which makes the joker chain live.
  * counter-hash policy: action = k-th valid action, k = hash(policy_seed, env_index, t) mod n_valid.
  * consumables that make the reference RAISE (The Hanged Man / Familiar / Grim / Incantation with a target card:
    `list.remove` of a class that is not in the deck; Sigil / Ouija: assignment to a frozen dataclass): a batch of envs
    cannot raise, so the step reports reward -1.0 with an error and the state stays exactly as the exception left it.
"""
from __future__ import annotations

import os
import random
import sys
import types

REFERENCE_ROOT = os.environ.get("BALATRO_REFERENCE", "/root/reference")


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "balatro_gym"))


def _install_gymnasium_shim():
    """Stand-in for the missing `gymnasium` package: space classes that only CARRY what they are given (shape, dtype, bounds,
    n) -- no arithmetic of the reference lives in gymnasium.  train_balatro_fixed.BalatroEnvFixed reads these
    attributes to build its fixed observation space, so they are kept faithfully."""
    if "gymnasium" in sys.modules:
        return
    import numpy as np
    gym = types.ModuleType("gymnasium")
    spaces = types.ModuleType("gymnasium.spaces")

    class Env:
        metadata = {}

        def __init__(self, *a, **k):
            pass

    class Wrapper(Env):
        def __init__(self, env):
            self.env = env
            self.action_space = getattr(env, "action_space", None)
            self.observation_space = getattr(env, "observation_space", None)

    class Space:
        def __init__(self, shape=None, dtype=None):
            self.shape = None if shape is None else tuple(shape)
            self.dtype = None if dtype is None else np.dtype(dtype)

    class Box(Space):
        def __init__(self, low, high, shape=None, dtype=np.float32):
            if shape is None:
                shape = np.shape(low)
            super().__init__(shape, dtype)
            self.low = np.broadcast_to(np.asarray(low), self.shape).astype(self.dtype) if np.ndim(low) else np.full(self.shape, low, dtype=self.dtype)
            self.high = np.broadcast_to(np.asarray(high), self.shape).astype(self.dtype) if np.ndim(high) else np.full(self.shape, high, dtype=self.dtype)

        def sample(self):
            return np.zeros(self.shape, dtype=self.dtype)

    class Discrete(Space):
        def __init__(self, n, *a, **k):
            super().__init__((), np.int64)
            self.n = int(n)

        def sample(self):
            return 0

    class MultiBinary(Space):
        def __init__(self, n, *a, **k):
            super().__init__((int(n),), np.int8)
            self.n = int(n)

    class Dict_(Space):
        def __init__(self, d=None, **k):
            super().__init__(None, None)
            self.spaces = dict(d or {}, **k)

    spaces.Space, spaces.Discrete, spaces.Box, spaces.MultiBinary, spaces.Dict = Space, Discrete, Box, MultiBinary, Dict_
    gym.Env, gym.Wrapper, gym.spaces = Env, Wrapper, spaces
    sys.modules["gymnasium"], sys.modules["gymnasium.spaces"] = gym, spaces


_loaded = {}


def load_reference():
    """Import the reference modules (once) and return them in a dict."""
    if _loaded:
        return _loaded
    if not reference_available():
        raise RuntimeError(f"reference not found at {REFERENCE_ROOT}")
    sys.dont_write_bytecode = True
    _install_gymnasium_shim()
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import balatro_gym.balatro_env_2 as env2
    import balatro_gym.unified_scoring as us
    import balatro_gym.scoring_engine as se
    import balatro_gym.complete_joker_effects as cje
    import balatro_gym.balatro_game as bg
    import balatro_gym.cards as cards
    import balatro_gym.jokers as jokers
    import balatro_gym.boss_blinds as bb

    # scorer-level joker semantics switch (see module docstring)
    orig_to_dict = env2.UnifiedGameState.to_dict

    def to_dict(self):
        d = orig_to_dict(self)
        if _loaded.get("_scorer_jokers_active"):
            d["jokers"] = [j.name for j in self.jokers]
        return d

    env2.UnifiedGameState.to_dict = to_dict
    _loaded.update(env2=env2, us=us, se=se, cje=cje, bg=bg, cards=cards, jokers=jokers, bb=bb)
    return _loaded


MASK64 = (1 << 64) - 1


def global_seed(seed: int) -> int:
    return (seed + 16000) % (2 ** 32)


def policy_hash(policy_seed: int, env_index: int, t: int) -> int:
    x = (policy_seed + 0x9E3779B97F4A7C15 * (env_index + 1) + 0xD1B54A32D192ED03 * (t + 1)) & MASK64
    x ^= x >> 30
    x = (x * 0xBF58476D1CE4E5B9) & MASK64
    x ^= x >> 27
    x = (x * 0x94D049BB133111EB) & MASK64
    x ^= x >> 31
    return x >> 32


POLICY_UNIFORM, POLICY_SMALL_ONLY, POLICY_CYCLE3 = 0, 1, 2


def policy_action(mask, phase: int, policy: int, policy_seed: int, env_index: int, t: int) -> int:
    if policy != POLICY_UNIFORM:
        if phase == 2:
            return 45 + (env_index % 3 if policy == POLICY_CYCLE3 else 0)
        if phase == 1:
            return 31
    valid = [a for a in range(60) if mask[a]]
    if not valid:
        return 0
    return valid[policy_hash(policy_seed, env_index, t) % len(valid)]


class RefEnv:
    """One reference env with its own 'global random' stream."""

    def __init__(self, seed: int, scorer_jokers: bool = False, max_ante: int = 0):
        self.ref = load_reference()
        self.scorer_jokers = scorer_jokers
        self.max_ante = max_ante
        random.seed(global_seed(seed))
        self._enter()
        try:
            self.env = self.ref["env2"].BalatroEnv(seed=seed)
        finally:
            self._leave()

    def _enter(self):
        self.ref["_scorer_jokers_active"] = self.scorer_jokers
        if hasattr(self, "_g"):
            random.setstate(self._g)

    def _leave(self):
        self._g = random.getstate()
        self.ref["_scorer_jokers_active"] = False

    def reset(self, seed=None):
        self._enter()
        try:
            return self.env.reset(seed=seed)[0]
        finally:
            self._leave()

    def obs(self):
        return self.env._get_observation()

    def step(self, action: int):
        self._enter()
        try:
            obs, reward, terminated, truncated, info = self.env.step(int(action))
        except (ValueError, AttributeError) as ex:  # dataclasses.FrozenInstanceError is an AttributeError
            if not (10 <= int(action) <= 14 and self.env.state.phase == 0):
                raise
            obs, reward, terminated, truncated = self.env._get_observation(), -1.0, False, False
            info = {"error": f"raised {type(ex).__name__}", "raised": True}
        finally:
            self._leave()
        if self.max_ante and self.env.state.ante > self.max_ante:
            terminated = True
            info["curriculum_limit_reached"] = True
        return obs, float(reward), bool(terminated), bool(truncated), info

    def set_jokers(self, ids):
        lib = {j.id: j for j in self.ref["jokers"].JOKER_LIBRARY}
        self.env.state.jokers = [lib[i] for i in ids]

    def set_card_state(self, deck_idx, enh=0, edi=0, seal=0):
        c = self.ref["cards"]
        self.env.state.card_states[deck_idx] = c.CardState(deck_idx, c.Enhancement(enh), c.Edition(edi), c.Seal(seal))

    def set_hand_level(self, hand_type, level):
        ht = self.ref["se"].HandType(hand_type)
        self.env.engine.set_hand_level(ht, level)
        self.env.state.hand_levels[ht] = self.env.engine.get_hand_level(ht)

    CONSUMABLE_NAMES = {
        1: 'The Fool', 2: 'The Magician', 3: 'The High Priestess', 4: 'The Empress', 5: 'The Emperor',
        6: 'The Hierophant', 7: 'The Lovers', 8: 'The Chariot', 9: 'Strength', 10: 'The Hermit', 11: 'Wheel of Fortune',
        12: 'Justice', 13: 'The Hanged Man', 14: 'Death', 15: 'Temperance', 16: 'The Devil', 17: 'The Tower',
        18: 'The Star', 19: 'The Moon', 20: 'The Sun', 21: 'Judgement', 22: 'The World',
        30: 'Mercury', 31: 'Venus', 32: 'Earth', 33: 'Mars', 34: 'Jupiter', 35: 'Saturn', 36: 'Uranus', 37: 'Neptune',
        38: 'Pluto', 39: 'Planet X', 40: 'Ceres', 41: 'Eris',
        50: 'Familiar', 51: 'Grim', 52: 'Incantation', 53: 'Talisman', 54: 'Aura', 55: 'Wraith', 56: 'Sigil', 57: 'Ouija',
        58: 'Ectoplasm', 59: 'Immolate', 60: 'Ankh', 61: 'Deja Vu', 62: 'Hex', 63: 'Trance', 64: 'Medium', 65: 'Cryptid',
        66: 'The Soul', 67: 'Black Hole'}

    def set_consumables(self, ids):
        """state.consumables by the ids of _get_consumable_ids (balatro_env_2.py:1545-1567)."""
        self.env.state.consumables = [self.CONSUMABLE_NAMES[i] for i in ids]

    def set_deck(self, codes):
        """state.deck / game.deck (ONE aliased list, balatro_env_2.py:528-531) re-ordered in place: codes are (rank-2)*4+suit."""
        c = self.ref["cards"]
        self.env.state.deck[:] = [c.Card(rank=c.Rank(v // 4 + 2), suit=c.Suit(v % 4)) for v in codes]
        assert self.env.game.deck is self.env.state.deck

    def set_money(self, money):
        self.env.state.money = money

    def set_ante(self, ante):
        self.env.state.ante = ante


# ---------------------------------------------------------------------------------------------------------
# balatro_sim.py: imports only with the reference's package directory itself on sys.path (bare
# `from scoring_engine import ...`, balatro_sim.py:6), and calculate_score calls a ScoreEngine.score that does not exist
# (:418; the result is only printed) -- SURVEY App. C step 4.
# ---------------------------------------------------------------------------------------------------------
SIM_SUITS = ["Clubs", "Diamonds", "Hearts", "Spades"]
SIM_ENH = [None, "bonus", "mult", "wild", "glass", "steel", "stone", "gold", "lucky"]
SIM_EDI = [None, "foil", "holographic", "polychrome", "negative"]
SIM_SEAL = [None, "gold", "red", "blue", "purple"]
SIM_TYPES = ["High Card", "Pair", "Two Pair", "Three of a Kind", "Straight", "Flush", "Full House", "Four of a Kind",
             "Straight Flush", "Five of a Kind", "Flush House", "Flush Five"]


def load_sim():
    if "sim" in _loaded:
        return _loaded["sim"]
    load_reference()
    pkg = os.path.join(REFERENCE_ROOT, "balatro_gym")
    if pkg not in sys.path:
        sys.path.insert(0, pkg)
    import balatro_gym.balatro_sim as sim
    _loaded["sim"] = sim
    return sim


def _sim_cards(sim, cards):
    return [sim.Card(rank=r, suit=SIM_SUITS[s], base_value=bv, enhancement=SIM_ENH[en], edition=SIM_EDI[ed], seal=SIM_SEAL[se])
            for (r, s, bv, en, ed, se) in cards]


def _utility_jokers(sim, s, four_fingers, shortcut):
    ids = {j.name: j.id for j in sim.JOKER_LIBRARY}
    assert ids["Four Fingers"] == 18 and ids["Shortcut"] == 69
    return ([18] if four_fingers else []) + ([69] if shortcut else [])


def sim_evaluate(cards, four_fingers=False, shortcut=False):
    """BalatroSimulator.evaluate_hand -> (top index, {type: (len(results[type]), positions of results[type][0])})."""
    sim = load_sim()
    s = sim.BalatroSimulator()
    s.player_state.jokers = _utility_jokers(sim, s, four_fingers, shortcut)
    objs = _sim_cards(sim, cards)
    res = s.evaluate_hand(objs)

    def positions(lst):
        return [next(i for i, o in enumerate(objs) if o is c) for c in lst]

    out = {}
    for t, name in enumerate(SIM_TYPES):
        r = res[name]
        out[t] = (len(r), positions(r[0]) if r else [])
    return SIM_TYPES.index(res["top"]), out


def sim_score(cards, jokers, game_state_keys, deck_len, seed):
    """BalatroSimulator.calculate_score after random.seed(seed).  jokers = player_state.jokers (ids; Four Fingers 18 / Shortcut 69
    among them change the evaluation and draw from the global stream like every other joker).  game_state_keys:
    None = the simulator's own _create_game_state() (no 'hands_left' / 'discards_left' keys -> the defaults 1 / 0 of
    complete_joker_effects.py:46-48), or a dict with hands_left / discards_left to add to it."""
    import contextlib
    import io
    sim = load_sim()
    s = sim.BalatroSimulator()
    s.score_engine.score = lambda ids, ht, lvl: s.score_engine.score_hand(ids, ht)
    s.player_state.jokers = list(jokers)
    s.player_state.deck = list(range(deck_len))
    gs = None
    if game_state_keys is not None:
        gs = s._create_game_state()
        gs.update(game_state_keys)
    random.seed(seed)
    with contextlib.redirect_stdout(io.StringIO()):
        score, state = s.calculate_score(_sim_cards(sim, cards), gs)
    probe = random.getrandbits(32)
    return int(score), int(state["money"]) - 100, probe


# ---------------------------------------------------------------------------------------------------------
# train_balatro_fixed.py: BalatroEnvFixed / SafeBalatroEnv, the wrappers every training script puts around the
# env.  The script imports stable_baselines3 (not installed: import-time only, no arithmetic -> empty stand-ins) and
# `balatro_gym.envs.balatro_env_2`, a path that does not exist in the tree (the env lives at balatro_gym/balatro_env_2.py):
# the module is registered under that name as well.
# ---------------------------------------------------------------------------------------------------------
def load_fixed_wrappers():
    if "fixed" in _loaded:
        return _loaded["fixed"]
    ref = load_reference()
    for name, attrs in (("stable_baselines3", ["PPO"]), ("stable_baselines3.common", []),
                        ("stable_baselines3.common.vec_env", ["SubprocVecEnv", "DummyVecEnv", "VecNormalize"]),
                        ("stable_baselines3.common.monitor", ["Monitor"]),
                        ("stable_baselines3.common.callbacks", ["CheckpointCallback", "BaseCallback", "EvalCallback"])):
        if name not in sys.modules:
            m = types.ModuleType(name)
            for a in attrs:
                setattr(m, a, type(a, (), {"__init__": lambda self, *x, **k: None}))
            sys.modules[name] = m
    if "balatro_gym.envs" not in sys.modules:
        pkg = types.ModuleType("balatro_gym.envs")
        pkg.__path__ = []
        sys.modules["balatro_gym.envs"] = pkg
    sys.modules["balatro_gym.envs.balatro_env_2"] = ref["env2"]
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        import train_balatro_fixed as tbf
    _loaded["fixed"] = tbf
    return tbf


class RefFixedEnv:
    """SafeBalatroEnv(BalatroEnvFixed(seed)) with this harness's per-env global stream (see RefEnv)."""

    def __init__(self, seed: int, max_invalid_actions: int = 50, max_episode_steps: int = 1000):
        import contextlib
        import io
        tbf = load_fixed_wrappers()
        random.seed(global_seed(seed))
        with contextlib.redirect_stdout(io.StringIO()):
            self.fixed = tbf.BalatroEnvFixed(seed=seed)
            self.env = tbf.SafeBalatroEnv(self.fixed, max_invalid_actions=max_invalid_actions, max_episode_steps=max_episode_steps)
        self._g = random.getstate()

    def _call(self, fn, *a, **k):
        import contextlib
        import io
        random.setstate(self._g)
        try:
            with contextlib.redirect_stdout(io.StringIO()):
                return fn(*a, **k)
        finally:
            self._g = random.getstate()

    def reset(self):
        return self._call(self.env.reset)[0]

    def step(self, action: int):
        return self._call(self.env.step, int(action))
