"""BalatroEnv -- single-env drop-in for `balatro_gym/balatro_env_2.py::BalatroEnv` (Gymnasium surface), backed by a
1-env BalatroVecEnv on the GPU.  Same constructor keywords, reset()/step() signatures, Discrete(60) action space,
observation keys and dtypes (numpy), reward float, info keys.  Meant for plumbing / parity checks (config 1 of
BASELINE.json); throughput comes from BalatroVecEnv.
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import Any, Dict, Optional

import numpy as np
import torch

from . import _native as nat
from .constants import ACTION_SPACE_SIZE, BOSS_BLIND_NAMES, HAND_TYPE_NAMES
from .vec_env import BalatroVecEnv

try:  # use gymnasium's spaces when it is installed, else minimal stand-ins with the same attributes
    import gymnasium as _gym
    from gymnasium import spaces as _spaces
    _EnvBase = _gym.Env
except Exception:  # pragma: no cover - gymnasium is not in this image
    class _Space:
        def __init__(self, low=None, high=None, shape=(), dtype=None):
            self.low, self.high, self.shape, self.dtype = low, high, tuple(shape), dtype

    class _Discrete(_Space):
        def __init__(self, n):
            super().__init__(0, n - 1, (), np.int64)
            self.n = int(n)

        def sample(self):
            return int(np.random.randint(self.n))

        def contains(self, x):
            return 0 <= int(x) < self.n

    class _MultiBinary(_Space):
        def __init__(self, n):
            super().__init__(0, 1, (n,), np.int8)
            self.n = n

    class _Dict(_Space):
        def __init__(self, d):
            super().__init__()
            self.spaces = dict(d)

        def __getitem__(self, k):
            return self.spaces[k]

        def keys(self):
            return self.spaces.keys()

    _spaces = SimpleNamespace(Box=_Space, Discrete=_Discrete, MultiBinary=_MultiBinary, Dict=_Dict)
    _EnvBase = object

# (low, high) of the produced observation keys, as declared by balatro_env_2.py:386-438
_BOUNDS = {
    "hand": (-1, 51), "hand_size": (0, 12), "deck_size": (0, 52), "chips_scored": (0, 10_000_000_000),
    "round_chips_scored": (0, 10_000_000), "progress_ratio": (0.0, 2.0), "mult": (0, 10_000),
    "chips_needed": (0, 10_000_000), "money": (-20, 999), "ante": (1, 1000), "round": (1, 3), "hands_left": (0, 12),
    "discards_left": (0, 10), "joker_count": (0, 10), "joker_ids": (0, 200), "joker_slots": (0, 10),
    "consumable_count": (0, 5), "consumables": (0, 100), "consumable_slots": (0, 5), "shop_items": (0, 300),
    "shop_costs": (0, 5000), "shop_rerolls": (0, 999), "hand_levels": (0, 15), "phase": (0, 3),
    "hands_played": (0, 10000), "best_hand_this_ante": (0, 10_000_000), "boss_blind_active": (0, 1),
    "boss_blind_type": (0, 30),
}
_ERRORS = {1: "Invalid action", 2: "Must play exactly 5 cards", 6: "Insufficient chips for reroll", 7: "Joker slots full",
           8: "Failed to use consumable"}
_TERMS = ["progress", "milestone", "score", "hand_quality", "efficiency", "synergy", "strategy", "ante_bonus"]


# The 20 keys balatro_env_2.py:439-468 DECLARES and `_get_observation` (:1473-1541) never produces: (low, high, shape, dtype).
# They are part of the space -- the reference's own wrapper iterates `env.observation_space.spaces.items()` (train_balatro_fixed.py:31)
# and zero-fills what is missing from an observation -- but never of an observation.
_DECLARED_ONLY = {
    "hand_one_hot": (0, 1, (8, 52), np.float32), "hand_suits": (0, 4, (8,), np.int8), "hand_ranks": (0, 13, (8,), np.int8),
    "rank_counts": (0, 4, (13,), np.int8), "suit_counts": (0, 8, (4,), np.int8), "straight_potential": (0, 1, (), np.float32),
    "flush_potential": (0, 1, (), np.float32), "avg_score_per_hand": (0, 10000, (), np.float32), "hands_until_shop": (0, 20, (), np.int8),
    "rounds_until_boss": (0, 3, (), np.int8), "has_mult_jokers": (0, 1, (), np.int8), "has_chip_jokers": (0, 1, (), np.int8),
    "has_xmult_jokers": (0, 1, (), np.int8), "has_economy_jokers": (0, 1, (), np.int8),
    "hand_potential_scores": (0, 10000, (12,), np.int32), "joker_synergy_score": (0, 10, (), np.float32),
    "risk_level": (0, 1, (), np.float32), "economy_health": (0, 1, (), np.float32), "blind_difficulty": (0, 1, (), np.float32),
    "win_probability": (0, 1, (), np.float32),
}


def make_observation_space():
    """The reference's `_create_observation_space` (balatro_env_2.py:386-470): all 51 declared keys in declaration order -- the 31
    `_get_observation` fills, then the 20 it never does."""
    d = {}
    for k in nat.OBS_KEYS:
        dt, shape = nat.OBS_SPEC[k]
        if k in ("selected_cards", "face_down_cards"):
            d[k] = _spaces.MultiBinary(8)
        elif k == "action_mask":
            d[k] = _spaces.MultiBinary(ACTION_SPACE_SIZE)
        else:
            lo, hi = _BOUNDS[k]
            d[k] = _spaces.Box(lo, hi, shape, dtype=np.dtype(dt).type)
    for k, (lo, hi, shape, dt) in _DECLARED_ONLY.items():
        d[k] = _spaces.Box(lo, hi, shape, dtype=dt)
    return _spaces.Dict(d)


class BalatroEnv(_EnvBase):
    metadata = {"render_modes": ["human", "rgb_array"], "render_fps": 4}

    def __init__(self, *, render_mode: str | None = None, seed: int | None = None, device: int = 0,
                 scorer_jokers: bool = False, max_ante: int = 0, card_states: bool = False):
        self.render_mode = render_mode
        self._seed = seed
        self.action_space = _spaces.Discrete(ACTION_SPACE_SIZE)
        self.observation_space = make_observation_space()
        self._vec = BalatroVecEnv(1, None if seed is None else [seed], device=device, scorer_jokers=scorer_jokers,
                                  autoreset=False, max_ante=max_ante, card_states=card_states,
                                  fused_steps=16)  # one step per call: shallow look-ahead rings (42 KB instead of 0.66 MB)
        self._action = torch.zeros(1, dtype=torch.int32, device=self._vec.device)
        # the scalars `state` reports, taken from the observation of the last reset() / step() -- a PRIVATE copy: the dict handed to the caller is the
        # caller's (a wrapper may edit it in place, train_balatro_fixed.py:125-207 does), `state` must keep saying what the env holds
        self._cached_obs: Optional[Dict[str, int]] = None

    # -- helpers
    def _np_obs(self) -> Dict[str, Any]:
        out = {}
        flat = self._vec.obs_flat.cpu().numpy()   # (one device-to-host copy of the 330 bytes; it synchronises the stream)
        for k in nat.OBS_KEYS:
            dt, shape = nat.OBS_SPEC[k]
            off, nb = self._vec._obs.layout[k]
            a = flat[off:off + nb].view(np.dtype(dt))
            out[k] = a.reshape(shape).copy() if shape else a.reshape(())[()]
        return out

    @property
    def state(self):
        """Read-only view of the scalars wrappers poke at (`env.state.ante` etc., train_balatro_agent.py:150)."""
        # (served from the observation the last reset() / step() / load_state() already brought to the host: the wrappers that poke at `env.state` do
        #  so between steps, and every access used to cost a device copy and a synchronisation of its own)
        o = self._cached_obs if self._cached_obs is not None else self._state_scalars(self._np_obs())
        return SimpleNamespace(**o)

    _STATE_KEYS = ("ante", "round", "money", "phase", "chips_needed", "chips_scored", "round_chips_scored", "hands_left", "discards_left",
                   "joker_slots", "hand_size")

    def _state_scalars(self, obs: Dict[str, Any]) -> Dict[str, int]:
        return {k: int(obs[k]) for k in self._STATE_KEYS}

    # -- gymnasium surface
    def reset(self, *, seed: int | None = None, options: dict | None = None):
        if seed is not None:
            self._seed = seed
            self._vec.reset(seed=[seed])
        else:
            self._vec.reset()
        obs = self._np_obs()
        self._cached_obs = self._state_scalars(obs)
        return obs, {}

    def step(self, action: int):
        self._action[0] = int(action)
        _, reward, term, trunc, info = self._vec.step(self._action)
        torch.cuda.synchronize(self._vec.device)
        r = float(reward.item())
        terminated = bool(term.item())
        inf = {k: v[0].cpu().numpy() for k, v in info.items()}
        out: Dict[str, Any] = {}
        err, flags, ht = int(inf["error"]), int(inf["flags"]), int(inf["hand_type"])
        if err in (9, 10):
            out["terminated"] = "max_ante_reached" if err == 9 else "max_score_reached"
        elif err in (3, 4, 5):  # the boss blind's own messages (boss_blinds.py:393,399,405); info.aux holds what they name
            aux = int(inf["aux"])
            out["error"] = (f"Cannot play {HAND_TYPE_NAMES[aux]} again" if err == 3 else
                            f"Can only play {HAND_TYPE_NAMES[aux]}" if err == 4 else f"Must play at least {aux} cards")
        elif err:
            out["error"] = _ERRORS.get(err, "error")
        if ht >= 0:
            terms = inf.get("reward_terms")
            if terms is not None:
                rb = {k: float(terms[i]) for i, k in enumerate(_TERMS)}
                out["reward_breakdown"] = rb
            bd = inf.get("score_breakdown")
            if bd is not None:  # balatro_env_2.py:909 (unified_scoring.py:129-137, :293-297); 'effects_applied' (display strings) is not produced
                fc, fm, fx, cc, bc, bm, mg = int(bd[0]), int(bd[1]), float(bd[2]), int(bd[3]), int(bd[4]), int(bd[5]), int(bd[6])
                out["score_breakdown"] = {"base_chips": bc, "base_mult": bm, "card_chips": cc, "joker_chips": fc - bc - cc,
                                          "joker_mult": fm - bm, "joker_x_mult": fx, "final_chips": fc, "final_mult": fm,
                                          "final_x_mult": fx, "final_score": int(fc * fm * fx), "money_gained": mg}
            out["final_score"] = int(inf["final_score"])
            out["hand_type"] = ht
            out["hand_type_name"] = HAND_TYPE_NAMES[ht]
            out["cards_played"] = int(inf["cards_played"])
        if flags & 1:
            out["beat_blind"] = True
        if flags & 2:
            out["failed"] = True
        if flags & 4:
            out["skipped_blind"] = True
        if flags & 8:
            out["opened_pack"] = True
        if flags & 16:
            out["bought_card"] = True
        if flags & 32:
            out["bought_voucher"] = ["Magic Trick", "Minimalist"][int(inf["aux"])]
        if flags & 128:
            out["sold_joker"] = int(inf["aux"])
        if flags & 256:
            out["curriculum_limit_reached"] = True
        if int(inf["aux"]) and not (flags & (8 | 32 | 64 | 128)) and not err and ht < 0:
            out["boss_blind"] = BOSS_BLIND_NAMES[int(inf["aux"])]
        obs = self._np_obs()
        self._cached_obs = self._state_scalars(obs)
        return obs, r, terminated, False, out

    def save_state(self):
        return {"blob": self._vec.get_state(0)}

    def load_state(self, saved):
        self._vec.set_state(0, saved["blob"])
        self._cached_obs = None

    def render(self):
        if self.render_mode != "human":
            return
        o = self._np_obs()
        print(f"Ante {int(o['ante'])} - Round {int(o['round'])} - Phase {int(o['phase'])} | "
              f"Score {int(o['round_chips_scored'])}/{int(o['chips_needed'])} | Money ${int(o['money'])} | "
              f"Hands {int(o['hands_left'])} Discards {int(o['discards_left'])} | Hand {o['hand'].tolist()}")

    def close(self):
        self._vec.close()

    # harness helper used by the parity tests
    def inject(self, **kw):
        self._vec.inject(**kw)
        self._cached_obs = None

    def inject_consumables(self, ids, apply_now: bool = True):
        """env.state.consumables = [names of ids] (ids as in balatro_env_2.py:1545-1567)."""
        self._vec.inject_consumables([list(ids)], apply_now=apply_now)
        self._cached_obs = None

    def inject_cards(self, cards, apply_now: bool = True):
        """cards = iterable of (deck_index, enhancement, edition, seal): env.card_states[idx] = CardState(...) (card_states=True)."""
        self._vec.inject_cards([list(cards)], apply_now=apply_now)
        self._cached_obs = None


def make_balatro_env(**kwargs):
    """Factory with the reference's signature (balatro_env_2.py:1803-1807)."""
    def _init():
        return BalatroEnv(**kwargs)
    return _init
