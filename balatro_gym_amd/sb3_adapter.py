"""SB3-style VecEnv over the MI355X env -- what the reference's training scripts wrap around `BalatroEnv`.

`hpc_train.py:60-65` / `train_balatro_fixed.py:285-300` build `SubprocVecEnv([Monitor(SafeBalatroEnv(BalatroEnvFixed(seed)))
for rank in range(n_envs)])`.  This class gives the same thing for N envs on one GPU, with the wrapper logic vectorised
on the device:

* `BalatroEnvFixed._fix_observation` (`train_balatro_fixed.py:125-207`): scalar Box -> shape (1,), MultiBinary -> int8 Box,
  int16/int8 array keys of its upgrade list -> int32, and the 20 keys the env declares (`balatro_env_2.py:386-470`) but
  `_get_observation()` never produces -> zeros of the declared shape / dtype.  `FIXED_SPEC` is that space.
* `SafeBalatroEnv` (`train_balatro_fixed.py:228-283`): 50 consecutive invalid actions (reward == -1.0, not done) ->
  terminated with reward -50.0 and `info['invalid_action_termination']`; 1000 steps in an episode -> truncated with
  `info['max_steps_reached']`.  (Its exception fallbacks have no counterpart: the device path does not raise.)
* `CurriculumBalatroEnv` (`train_balatro_agent.py:146-152`): `max_ante` of the underlying env.
* `stable_baselines3.common.vec_env.VecEnv` conventions: `reset() -> obs`, `step_async(actions)` / `step_wait() ->
  (obs, rewards float32, dones bool, infos)`; finished envs are reset in the same call, their `infos[i]` carries
  `terminal_observation` (the observation before the reset; for a game-over the env's SAME_STEP auto-reset has already
  replaced it, so it is only present for wrapper-made endings) and `TimeLimit.truncated`.

SB3 itself is not a dependency (it is not installed in the build image): the class is a duck type of VecEnv and becomes a
real subclass when `stable_baselines3` is importable.  Observations leave as numpy arrays (SB3's contract) or, with
`as_torch=True`, as torch tensors on the env's device.
"""
from __future__ import annotations

from typing import Any, Dict, List, Optional, Sequence

import numpy as np
import torch

from . import _native as nat
from .vec_env import BalatroVecEnv

try:  # pragma: no cover - SB3 is optional
    from stable_baselines3.common.vec_env import VecEnv as _SB3VecEnv
except Exception:  # noqa: BLE001
    _SB3VecEnv = object

_NP2T = {"int8": torch.int8, "int16": torch.int16, "int32": torch.int32, "int64": torch.int64, "float32": torch.float32}

# key -> (dtype, shape, produced).  Order = the reference's observation_space (balatro_env_2.py:388-468).
_UPGRADE = {"chips_scored", "round_chips_scored", "chips_needed", "shop_costs", "shop_items", "joker_ids", "shop_rerolls",
            "hand_potential_scores", "best_hand_this_ante", "money", "hands_played", "ante", "round_chips_scored_rank"}
_MULTIBINARY = {"selected_cards": 8, "action_mask": 60, "face_down_cards": 8}
_NEVER_PRODUCED = {  # declared Box(shape, dtype) the env never fills (zero-filled by BalatroEnvFixed)
    "hand_one_hot": ("float32", (8, 52)), "hand_suits": ("int8", (8,)), "hand_ranks": ("int8", (8,)),
    "rank_counts": ("int8", (13,)), "suit_counts": ("int8", (4,)), "straight_potential": ("float32", ()),
    "flush_potential": ("float32", ()), "avg_score_per_hand": ("float32", ()), "hands_until_shop": ("int8", ()),
    "rounds_until_boss": ("int8", ()), "has_mult_jokers": ("int8", ()), "has_chip_jokers": ("int8", ()),
    "has_xmult_jokers": ("int8", ()), "has_economy_jokers": ("int8", ()), "hand_potential_scores": ("int32", (12,)),
    "joker_synergy_score": ("float32", ()), "risk_level": ("float32", ()), "economy_health": ("float32", ()),
    "blind_difficulty": ("float32", ()), "win_probability": ("float32", ()),
}


def _fixed_spec() -> Dict[str, tuple]:
    spec: Dict[str, tuple] = {}
    declared = dict(nat.OBS_SPEC)
    for k in nat.OBS_KEYS:
        dt, shape = declared[k]
        if k in _MULTIBINARY:                       # MultiBinary(n) -> Box(0, 1, (n,), int8)
            spec[k] = ("int8", (_MULTIBINARY[k],), True)
        elif shape == ():                           # scalar Box -> (1,), dtype kept
            spec[k] = (dt, (1,), True)
        elif dt in ("int16", "int8") and k in _UPGRADE:  # int upgrade of array keys
            spec[k] = ("int32", shape, True)
        else:
            spec[k] = (dt, shape, True)
    for k, (dt, shape) in _NEVER_PRODUCED.items():
        spec[k] = (dt, (1,) if shape == () else shape, False)
    return spec


FIXED_SPEC = _fixed_spec()


def fix_observation(obs: Dict[str, torch.Tensor], zeros: Optional[Dict[str, torch.Tensor]] = None) -> Dict[str, torch.Tensor]:
    """Batched `_fix_observation`: obs[k] is [N, ...] in the env's dtypes; returns the 51-key dict of FIXED_SPEC."""
    n = next(iter(obs.values())).shape[0]
    dev = next(iter(obs.values())).device
    out: Dict[str, torch.Tensor] = {}
    for k, (dt, shape, produced) in FIXED_SPEC.items():
        if not produced:
            out[k] = zeros[k] if zeros is not None else torch.zeros((n,) + shape, dtype=_NP2T[dt], device=dev)
            continue
        v = obs[k]
        if v.dim() == 1:
            v = v.reshape(n, 1)
        if dt == "int32" and v.dtype == torch.int64:  # np.clip(value, int32 min, int32 max) of the int upgrade
            v = v.clamp(-2 ** 31, 2 ** 31 - 1)
        out[k] = v.to(_NP2T[dt])
    return out


class BalatroSB3VecEnv(_SB3VecEnv):
    """N Balatro envs on one GPU behind the VecEnv calling convention (see module docstring)."""

    def __init__(self, num_envs: int, seed: int = 0, *, device: int = 0, max_invalid_actions: int = 50,
                 max_episode_steps: int = 1000, max_ante: int = 0, scorer_jokers: bool = False, as_torch: bool = False,
                 seeds: Optional[Sequence[int]] = None):
        self.num_envs = int(num_envs)
        self.as_torch = bool(as_torch)
        self.max_invalid_actions = int(max_invalid_actions)
        self.max_episode_steps = int(max_episode_steps)
        # make_env_fixed(seed, rank): BalatroEnvFixed(seed=seed + rank) (train_balatro_fixed.py:285-288)
        self.seeds = list(seeds) if seeds is not None else [seed + r for r in range(self.num_envs)]
        self.env = BalatroVecEnv(self.num_envs, self.seeds, device=device, scorer_jokers=scorer_jokers, autoreset=True,
                                 max_ante=max_ante, fused_steps=32)  # a VecEnv steps once per call: shallow look-ahead rings
        dev = self.env.device
        self.observation_spec = FIXED_SPEC
        self.num_actions = 60
        self._zeros = {k: torch.zeros((self.num_envs,) + shape, dtype=_NP2T[dt], device=dev)
                       for k, (dt, shape, produced) in FIXED_SPEC.items() if not produced}
        self._invalid = torch.zeros(self.num_envs, dtype=torch.int32, device=dev)
        self._steps = torch.zeros(self.num_envs, dtype=torch.int32, device=dev)
        self._actions: Optional[torch.Tensor] = None
        self._make_spaces()

    def _make_spaces(self):
        try:  # gymnasium is optional too
            from gymnasium import spaces
            d = {}
            for k, (dt, shape, _) in FIXED_SPEC.items():
                info = np.iinfo(dt) if dt.startswith("int") else np.finfo(dt)
                d[k] = spaces.Box(low=info.min, high=info.max, shape=shape, dtype=np.dtype(dt))
            self.observation_space = spaces.Dict(d)
            self.action_space = spaces.Discrete(60)
        except Exception:  # noqa: BLE001
            self.observation_space = {k: (np.dtype(dt), shape) for k, (dt, shape, _) in FIXED_SPEC.items()}
            self.action_space = 60

    # ---- VecEnv calling convention ----------------------------------------------------------------------------
    def _out(self, obs: Dict[str, torch.Tensor]):
        fixed = fix_observation(obs, self._zeros)
        if self.as_torch:
            return {k: v.clone() for k, v in fixed.items()}
        return {k: v.cpu().numpy() for k, v in fixed.items()}

    def reset(self):
        self._invalid.zero_()
        self._steps.zero_()
        return self._out(self.env.reset())

    def step_async(self, actions):
        a = actions if isinstance(actions, torch.Tensor) else torch.as_tensor(np.asarray(actions))
        self._actions = a.to(device=self.env.device, dtype=torch.int32).reshape(self.num_envs).contiguous()

    def step_wait(self):
        assert self._actions is not None, "step_async first"
        obs, reward, term, trunc, _info = self.env.step(self._actions)
        self._actions = None
        reward = reward.clone()
        env_term = term.to(torch.bool).clone()  # game over: the device env has already started the next episode
        self._steps += 1
        # SafeBalatroEnv.step (train_balatro_fixed.py:239-260)
        invalid = (reward == -1.0) & ~env_term
        self._invalid = torch.where(invalid, self._invalid + 1, torch.zeros_like(self._invalid))
        kill = invalid & (self._invalid >= self.max_invalid_actions)
        reward = torch.where(kill, torch.full_like(reward, -50.0), reward)
        truncated = self._steps >= self.max_episode_steps
        wrapper_end = (kill | truncated) & ~env_term  # episodes the WRAPPER ended: these envs still need their reset
        done = env_term | kill | truncated
        terminal_obs = None
        if bool(wrapper_end.any()):
            idx = wrapper_end.nonzero(as_tuple=False).flatten()
            fixed_before = fix_observation(obs, self._zeros)
            terminal_obs = (idx.cpu().numpy(), {k: v[idx].cpu().numpy() for k, v in fixed_before.items()})
            obs = self.env.reset(mask=wrapper_end)
        self._steps = torch.where(done, torch.zeros_like(self._steps), self._steps)
        self._invalid = torch.where(done, torch.zeros_like(self._invalid), self._invalid)
        # infos: SB3 wants one dict per env; only finished envs carry anything
        infos: List[Dict[str, Any]] = [{} for _ in range(self.num_envs)]
        done_h, kill_h, trunc_h = done.cpu().numpy(), kill.cpu().numpy(), truncated.cpu().numpy()
        for i in np.flatnonzero(done_h):
            infos[i]["TimeLimit.truncated"] = bool(trunc_h[i] and not kill_h[i])
            if kill_h[i]:
                infos[i]["invalid_action_termination"] = True
            if trunc_h[i]:
                infos[i]["max_steps_reached"] = True
        if terminal_obs is not None:
            rows, tob = terminal_obs
            for j, i in enumerate(rows):
                infos[int(i)]["terminal_observation"] = {k: v[j] for k, v in tob.items()}
        out = self._out(obs)
        if self.as_torch:
            return out, reward.to(torch.float32), done, infos
        return out, reward.to(torch.float32).cpu().numpy(), done_h.astype(bool), infos

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def close(self):
        self.env.close()

    # the rest of the VecEnv surface, as far as it makes sense for envs that are not Python objects
    def seed(self, seed: Optional[int] = None):
        if seed is not None:
            self.seeds = [seed + r for r in range(self.num_envs)]
            self.env.seed(self.seeds)
        return list(self.seeds)

    def get_attr(self, attr_name: str, indices=None):
        n = self.num_envs if indices is None else len(list(indices))
        return [getattr(self, attr_name)] * n

    def set_attr(self, attr_name: str, value, indices=None):
        setattr(self, attr_name, value)

    def env_method(self, method_name: str, *args, indices=None, **kwargs):
        raise NotImplementedError("the envs of BalatroSB3VecEnv live on the GPU; there are no per-env Python methods")

    def env_is_wrapped(self, wrapper_class, indices=None):
        n = self.num_envs if indices is None else len(list(indices))
        return [False] * n


class CurriculumTracker:
    """`CurriculumBalatroEnv` (train_balatro_agent.py:126-170), vectorised: one `current_max_ante` per env, raised by
    `ante_increment` when the env has played at least 100 episodes at its level and `success_threshold` of its last 100
    episodes ended at an ante >= the cap.  The cap itself lives in the device env (bg_set_max_ante: episodes end with
    `curriculum_limit_reached` as soon as ante > cap); this class only keeps the per-env episode statistics the rule needs and
    decides WHEN to raise a cap.  Call `update(done, ante_at_end)` after every step with the finished envs' final antes."""

    def __init__(self, env: BalatroVecEnv, initial_max_ante: int = 3, ante_increment: int = 1, success_threshold: float = 0.8,
                 window: int = 100):
        self.env = env
        n = env.num_envs
        self.cap = np.full(n, int(initial_max_ante), np.int32)
        self.inc, self.thr, self.window = int(ante_increment), float(success_threshold), int(window)
        self.hist = np.zeros((n, self.window), np.int32)   # final antes of the last `window` episodes
        self.count = np.zeros(n, np.int64)                 # episodes ever finished
        self.at_level = np.zeros(n, np.int64)              # episodes at the current level (`episodes_at_current_level`)
        env.set_max_ante(self.cap)

    def update(self, done: np.ndarray, ante_at_end: np.ndarray) -> np.ndarray:
        """done: bool [N]; ante_at_end: int [N] (state.ante when the episode ended).  Returns the mask of envs whose cap rose."""
        done = np.asarray(done, bool)
        idx = np.flatnonzero(done)
        if idx.size == 0:
            return np.zeros(len(done), bool)
        self.hist[idx, self.count[idx] % self.window] = np.asarray(ante_at_end)[idx]
        self.count[idx] += 1
        self.at_level[idx] += 1  # the reference counts in reset(); one reset follows every finished episode
        raised = np.zeros(len(done), bool)
        for i in idx:  # train_balatro_agent.py:157-166
            if self.at_level[i] >= self.window:
                recent = self.hist[i] if self.count[i] >= self.window else self.hist[i, :self.count[i]]
                if (recent >= self.cap[i]).sum() / float(self.window) >= self.thr:
                    self.cap[i] += self.inc
                    self.at_level[i] = 0
                    raised[i] = True
        if raised.any():
            self.env.set_max_ante(self.cap, mask=raised)
        return raised
