"""BalatroVecEnv -- N lockstep Balatro envs on one MI355X behind the reference's reset()/step() surface.

Host-side mirror of `balatro_gym/balatro_env_2.py::BalatroEnv` (observation keys/dtypes, Discrete(60) actions, reward
/ terminated / truncated / info semantics), vectorised: every quantity gains a leading [N] axis and lives in HBM as a
torch tensor.  All game logic runs in libbalatro_mi355x.so (hand-written HIP); torch only owns device buffers and the
stream.  There is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import random as _pyrandom
from typing import Dict, Optional, Sequence

import numpy as np
import torch

from . import _native as nat

_TORCH_DT = {"int8": torch.int8, "int16": torch.int16, "int32": torch.int32, "int64": torch.int64,
             "float32": torch.float32, "float64": torch.float64, "uint8": torch.uint8}


def _align(n: int, a: int = 256) -> int:
    return (n + a - 1) // a * a


def obs_flat_bytes(n: int, steps: int = 1) -> int:
    """Bytes of the flat observation buffer of `n` envs (every key's array, each aligned like `ObsBuffers` lays them out)."""
    off = 0
    for k in nat.OBS_KEYS:
        dt, shape = nat.OBS_SPEC[k]
        off = _align(off + n * steps * int(np.prod(shape, dtype=np.int64)) * np.dtype(dt).itemsize)
    return off


class ObsBuffers:
    """The 31 observation arrays as views into ONE flat device byte buffer (one collective gathers all keys)."""

    def __init__(self, n: int, device: torch.device, steps: int = 1):
        self.n, self.steps = n, steps
        rows = n * steps
        off = 0
        self.layout = {}
        for k in nat.OBS_KEYS:
            dt, shape = nat.OBS_SPEC[k]
            nbytes = rows * int(np.prod(shape, dtype=np.int64)) * np.dtype(dt).itemsize
            self.layout[k] = (off, nbytes)
            off = _align(off + nbytes)
        self.flat = torch.zeros(off, dtype=torch.uint8, device=device)
        self.tensors: Dict[str, torch.Tensor] = {}
        for k in nat.OBS_KEYS:
            dt, shape = nat.OBS_SPEC[k]
            o, nb = self.layout[k]
            lead = (steps, n) if steps > 1 else (n,)
            self.tensors[k] = self.flat[o:o + nb].view(_TORCH_DT[dt]).view(*lead, *shape)
        self.ptrs = nat.ObsPtrs(**{k: self.tensors[k].data_ptr() for k in nat.OBS_KEYS})


class RowBuffers:
    """[steps, N] packed records for `BalatroVecEnv.rollout` (bg_rollout_rows): one 352-byte record per (step, env),
    every observation key -- plus the step's reward / action / terminated -- a strided, correctly typed VIEW of the
    same byte tensor (`tensors[key]`, `reward`, `action`, `terminated`); `.contiguous()` gives the dense per-key array.
    """

    def __init__(self, n: int, device: torch.device, steps: int = 1, row_stride: int = 0):
        """row_stride: bytes from one record to the next (a multiple of 16, >= 352; 0 = 352, densely packed).  384
        (`_native.ROW_STRIDE_LINES`) is the FAST layout: every record is written as three whole 128-byte lines (bytes 352..383 zeros),
        which the HBM takes 1.7x faster than records that end in partial lines."""
        self.n, self.steps = n, steps
        self.row_stride = int(row_stride) or nat.ROW_BYTES
        if self.row_stride < nat.ROW_BYTES or self.row_stride % 16:
            raise ValueError("row_stride must be a multiple of 16 and >= the 352-byte record")
        self.rows = torch.zeros((steps, n, self.row_stride), dtype=torch.uint8, device=device)
        self.tensors: Dict[str, torch.Tensor] = {}
        for k in nat.OBS_KEYS:
            dt, shape = nat.OBS_SPEC[k]
            self.tensors[k] = self._view(nat.ROW_OFFSETS[k], dt, shape)
        self.reward = self._view(nat.ROW_EXTRA["reward"][0], "float64", ())
        self.action = self._view(nat.ROW_EXTRA["action"][0], "int32", ())
        self.terminated = self._view(nat.ROW_EXTRA["terminated"][0], "uint8", ())

    def _view(self, off: int, dt: str, shape) -> torch.Tensor:
        item = np.dtype(dt).itemsize
        count = int(np.prod(shape, dtype=np.int64))
        v = self.rows[:, :, off:off + count * item].view(_TORCH_DT[dt])  # [steps, n, count], last dim contiguous
        return v if shape else v[:, :, 0]


class BalatroVecEnv:
    """N independent Balatro games stepped in lockstep on one GPU.

    seeds[i] plays the role of `BalatroEnv(seed=seeds[i])` (balatro_env_2.py:359): it seeds the 16 named streams
    (`(seed + 1000*i) % 2**32`, :105) and -- harness convention -- a per-env stand-in for the process-global
    `random` module with G(seed) = (seed + 16000) % 2**32.  Like the reference constructor, __init__ ends with reset().
    """

    num_actions = 60

    def __init__(self, num_envs: int, seeds: Optional[Sequence[int]] = None, *, device: int | str | torch.device = 0,
                 scorer_jokers: bool = False, autoreset: bool = True, max_ante: int = 0, info_terms: bool = True,
                 card_states: bool = False, fused_steps: int = 0, obs_layout: str = "keys"):
        """obs_layout: "keys" -- one contiguous tensor per observation key (bg_step / bg_observe, the reference's dict of arrays);
        "rows" -- ONE packed 384-byte record per env (bg_step_rows / bg_observe_rows): `obs[key]` are strided, correctly typed views of it
        (`obs_rows` is the [N, 384] byte tensor, what a policy network would concatenate anyway) and `step()` is ~10 % shorter (bench.py `step_path`)."""
        if obs_layout not in ("keys", "rows"):
            raise ValueError("obs_layout must be 'keys' or 'rows'")
        self.obs_layout = obs_layout
        if not torch.cuda.is_available():
            raise nat.NativeError("BalatroVecEnv needs a HIP device (torch.cuda.is_available() is False); "
                                  "there is no CPU fallback")
        self._L = nat.load()
        self.device = torch.device(device if not isinstance(device, int) else f"cuda:{device}")
        self.num_envs = int(num_envs)
        self.autoreset = bool(autoreset)
        self.scorer_jokers = bool(scorer_jokers)
        self.max_ante = int(max_ante)
        self.card_states = bool(card_states)
        flags = ((nat.FLAG_SCORER_JOKERS if scorer_jokers else 0) | (nat.FLAG_AUTORESET if autoreset else 0) |
                 (nat.FLAG_CARD_STATES if card_states else 0))
        self._h = C.c_void_p()
        # fused_steps: the longest rollout / step_many the caller will fuse into one launch (0 = 372-step launches): sizes the RNG
        # look-ahead rings, i.e. the HBM this handle holds (0.66 MB per env at full depth, ~42 KB per env at 16)
        rc = self._L.bg_create_ex(self.num_envs, self.device.index or 0, flags, self.max_ante, int(fused_steps), C.byref(self._h))
        if rc != 0:
            raise nat.NativeError(f"bg_create failed ({rc}): {self._L.bg_last_error(None).decode()}")
        n, dev = self.num_envs, self.device
        with torch.cuda.device(dev):
            self._obs = ObsBuffers(n, dev)
            self._rowbuf = RowBuffers(n, dev, steps=1, row_stride=nat.ROW_STRIDE_LINES) if obs_layout == "rows" else None
            self._row_tensors = {k: v[0] for k, v in self._rowbuf.tensors.items()} if self._rowbuf is not None else None
            self.reward = torch.zeros(n, dtype=torch.float64, device=dev)
            self.terminated = torch.zeros(n, dtype=torch.uint8, device=dev)
            self.truncated = torch.zeros(n, dtype=torch.uint8, device=dev)
            self.info = {k: torch.zeros((n,) + nat.INFO_SPEC[k][1], dtype=_TORCH_DT[nat.INFO_SPEC[k][0]], device=dev)
                         for k in nat.INFO_KEYS if info_terms or k not in ("reward_terms", "score_breakdown")}
            self._info_ptrs = nat.InfoPtrs(**{k: (self.info[k].data_ptr() if k in self.info else None)
                                              for k in nat.INFO_KEYS})
            self._stats = torch.zeros(6, dtype=torch.int64, device=dev)
        # the constant arguments of bg_step, converted once
        self._step_args = (C.byref(self._obs.ptrs), C.c_void_p(self.reward.data_ptr()), C.c_void_p(self.terminated.data_ptr()),
                           C.c_void_p(self.truncated.data_ptr()), C.byref(self._info_ptrs))
        if self._rowbuf is not None:
            self._rows_args = (C.c_void_p(self._rowbuf.rows.data_ptr()), C.c_uint64(self._rowbuf.row_stride))
        self.seed(seeds)
        self.reset()

    # ------------------------------------------------------------------ plumbing
    def _check(self, rc: int, what: str):
        if rc != 0:
            raise nat.NativeError(f"{what} failed ({rc}): {self._L.bg_last_error(self._h).decode()}")

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    @property
    def obs(self) -> Dict[str, torch.Tensor]:
        """The live observation tensors (updated in place by reset/step; clone() to keep a copy).  obs_layout "rows": views of `obs_rows`."""
        return self._row_tensors if self._row_tensors is not None else self._obs.tensors

    @property
    def obs_rows(self) -> torch.Tensor:
        """obs_layout "rows": the [N, 384] byte tensor of packed records (BG_ROW_* offsets) behind `obs`."""
        if self._rowbuf is None:
            raise AttributeError("obs_rows exists with obs_layout='rows'")
        return self._rowbuf.rows[0]

    @property
    def obs_flat(self) -> torch.Tensor:
        return self._rowbuf.rows[0].reshape(-1) if self._rowbuf is not None else self._obs.flat

    def set_gather_peers(self, buffers: Sequence[Optional[torch.Tensor]], rank: int) -> None:
        """Sharded jobs: `buffers[r]` = rank r's gather buffer (uint8 [world, N, 352]) as a tensor in THIS process -- the own one allocated here, the
        peers' opened from their CUDA IPC handles.  From now on the last launch of every packed-record `rollout` also writes every env's current
        record into slot [rank] of all of them (bg_set_gather_peers).  An empty list switches it off."""
        world = len(buffers)
        if world == 0:
            self._check(self._L.bg_set_gather_peers(self._h, None, 0, 0), "bg_set_gather_peers")
            self._gather_keep = None
            return
        for b in buffers:
            if b.dtype != torch.uint8 or tuple(b.shape) != (world, self.num_envs, nat.ROW_BYTES) or not b.is_contiguous():
                raise ValueError(f"every gather buffer must be a contiguous uint8 [{world}, {self.num_envs}, {nat.ROW_BYTES}] tensor")
        arr = (C.c_void_p * world)(*[b.data_ptr() for b in buffers])
        self._check(self._L.bg_set_gather_peers(self._h, C.cast(arr, C.c_void_p), world, int(rank)), "bg_set_gather_peers")
        self._gather_keep = list(buffers)   # the mappings must outlive the handle's use of them

    def obs_flat_bytes(self, n: int) -> int:
        """Bytes `obs_flat` takes for `n` envs in THIS env's layout: n records of `row_stride` bytes with obs_layout "rows", else the per-key
        arrays as `ObsBuffers` lays them out (what a sharded gather pads every rank's buffer to)."""
        return int(n) * self._rowbuf.row_stride if self._rowbuf is not None else obs_flat_bytes(int(n))

    def state_bytes(self) -> int:
        return int(self._L.bg_state_bytes(self._h))

    @property
    def max_fused_steps(self) -> int:
        """Steps `rollout` runs as one kernel launch (bg_max_fused_steps): the natural length of [T, N] obs buffers."""
        return int(self._L.bg_max_fused_steps(self._h))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            torch.cuda.synchronize(self.device)
            self._L.bg_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ reference surface
    def seed(self, seeds: Optional[Sequence[int]] = None, mask: Optional[Sequence[bool]] = None, reseed_global: bool = True):
        """`DeterministicRNG(seed)` per env (balatro_env_2.py:84-106).  None -> `random.randint(0, 2**32-1)` (:88)."""
        n = self.num_envs
        if seeds is None:
            seeds = [_pyrandom.randint(0, 2 ** 32 - 1) for _ in range(n)]
        arr = np.ascontiguousarray(np.asarray(seeds, dtype=np.int64))
        if arr.shape != (n,):
            raise ValueError(f"seeds must have shape ({n},)")
        m = None if mask is None else np.ascontiguousarray(np.asarray(mask, dtype=np.uint8))
        with torch.cuda.device(self.device):
            self._check(self._L.bg_seed(self._h, arr.ctypes.data_as(C.c_void_p),
                                        None if m is None else m.ctypes.data_as(C.c_void_p),
                                        1 if reseed_global else 0, self._stream()), "bg_seed")
        self.seeds = arr.copy() if mask is None else np.where(np.asarray(mask, bool), arr, getattr(self, "seeds", arr))

    def reset(self, *, seed: Optional[Sequence[int]] = None, mask: Optional[torch.Tensor] = None):
        """`reset(seed=...)` for the masked envs (all when mask is None); returns the observation dict of ALL envs."""
        if seed is not None:
            hm = None if mask is None else mask.to("cpu").numpy().astype(np.uint8)
            self.seed(seed, hm, reseed_global=False)  # reset(seed=s) rebuilds the streams, not the global module (:507-509)
        mptr = None
        if mask is not None:
            mask = mask.to(device=self.device, dtype=torch.uint8).contiguous()
            mptr = C.c_void_p(mask.data_ptr())
        with torch.cuda.device(self.device):
            if self._rowbuf is not None:
                self._check(self._L.bg_reset(self._h, mptr, None, self._stream()), "bg_reset")
                self._check(self._L.bg_observe_rows(self._h, self._rows_args[0], self._rows_args[1], self._stream()), "bg_observe_rows")
            else:
                self._check(self._L.bg_reset(self._h, mptr, C.byref(self._obs.ptrs), self._stream()), "bg_reset")
        return self.obs

    def step(self, actions: torch.Tensor):
        """One lockstep `step(action)` (balatro_env_2.py:616).  actions: int32 [N] on this device."""
        if actions.dtype != torch.int32 or actions.device != self.device or not actions.is_contiguous():
            actions = actions.to(device=self.device, dtype=torch.int32).contiguous()
        # (no torch.cuda.device() context: every entry point of the library switches to the handle's device itself, and a step is
        #  short enough for two extra hipSetDevice calls and seven ctypes conversions to show)
        a = self._step_args
        if self._rowbuf is not None:
            r = self._rows_args
            rc = self._L.bg_step_rows(self._h, actions.data_ptr(), r[0], r[1], a[1], a[2], a[3], a[4],
                                      torch.cuda.current_stream(self.device).cuda_stream)
            if rc != 0:
                self._check(rc, "bg_step_rows")
            return self._row_tensors, self.reward, self.terminated, self.truncated, self.info
        rc = self._L.bg_step(self._h, actions.data_ptr(), a[0], a[1], a[2], a[3], a[4],
                             torch.cuda.current_stream(self.device).cuda_stream)
        if rc != 0:
            self._check(rc, "bg_step")
        return self._obs.tensors, self.reward, self.terminated, self.truncated, self.info

    def step_many(self, actions: torch.Tensor, obs_buffers: Optional["ObsBuffers"] = None, reward: Optional[torch.Tensor] = None,
                  terminated: Optional[torch.Tensor] = None):
        """K consecutive `step()` calls in one launch (bg_step_many): actions int32 [K, N].  With `obs_buffers` of K rows
        (and optional [K, N] reward / terminated tensors) every call's outputs are kept; otherwise the live tensors hold the last
        call's observation / reward / terminated / info."""
        if actions.dtype != torch.int32 or actions.device != self.device or not actions.is_contiguous():
            actions = actions.to(device=self.device, dtype=torch.int32).contiguous()
        if actions.dim() != 2 or actions.shape[1] != self.num_envs:
            raise ValueError(f"actions must be [K, {self.num_envs}]")
        K = int(actions.shape[0])
        keep = obs_buffers is not None and obs_buffers.steps > 1
        if keep and obs_buffers.steps < K:
            raise ValueError("obs_buffers has fewer rows than steps")
        ob = obs_buffers if keep else self._obs
        rw = reward if (keep and reward is not None) else self.reward
        tm = terminated if (keep and terminated is not None) else self.terminated
        with torch.cuda.device(self.device):
            self._check(self._L.bg_step_many(
                self._h, K, C.c_void_p(actions.data_ptr()), C.byref(ob.ptrs), 1 if keep else 0, C.c_void_p(rw.data_ptr()),
                C.c_void_p(tm.data_ptr()), None if keep else C.c_void_p(self.truncated.data_ptr()),
                None if keep else C.byref(self._info_ptrs), self._stream()), "bg_step_many")
        if self._rowbuf is not None:
            self.observe()   # (obs_layout "rows": the live records follow)
            if not keep:
                return self._row_tensors, rw, tm, self.truncated, self.info
        return ob.tensors, rw, tm, self.truncated, self.info

    def observe(self):
        with torch.cuda.device(self.device):
            if self._rowbuf is not None:
                self._check(self._L.bg_observe_rows(self._h, self._rows_args[0], self._rows_args[1], self._stream()), "bg_observe_rows")
            else:
                self._check(self._L.bg_observe(self._h, C.byref(self._obs.ptrs), self._stream()), "bg_observe")
        return self.obs

    # ------------------------------------------------------------------ extras
    def rollout(self, steps: int, policy: int = nat.POLICY_UNIFORM, policy_seed: int = 0, env_index0: int = 0,
                t0: int = 0, obs_buffers: Optional[ObsBuffers] = None, reward: Optional[torch.Tensor] = None,
                terminated: Optional[torch.Tensor] = None, actions: Optional[torch.Tensor] = None,
                zero_stats: bool = True):
        """Fused random-policy rollout (bg_rollout).  With obs_buffers of `steps` rows every step's observation is
        kept ([T, N, ...]); otherwise the live observation tensors are overwritten each step.  A `RowBuffers` selects
        the packed-record output (bg_rollout_rows): same values, one record per (step, env)."""
        if isinstance(obs_buffers, RowBuffers):
            if obs_buffers.steps < steps and obs_buffers.steps > 1:
                raise ValueError("obs_buffers has fewer rows than steps")
            if reward is not None or terminated is not None or actions is not None:
                raise ValueError("packed records already carry reward / action / terminated")
            if zero_stats:
                self._stats.zero_()
            # (no torch.cuda.device() context, as in step(): the library switches to the handle's device itself, and on a 20-step launch the
            #  two device exchanges of the context manager are a few per cent of the call)
            rc = self._L.bg_rollout_rows(
                self._h, int(steps), int(policy), C.c_uint64(policy_seed), C.c_uint64(env_index0), C.c_uint64(t0),
                C.c_void_p(obs_buffers.rows.data_ptr()), C.c_uint64(getattr(obs_buffers, "row_stride", nat.ROW_BYTES)), 1 if obs_buffers.steps > 1 else 0,
                C.c_void_p(self._stats.data_ptr()), self._stream())
            if rc != 0:
                self._check(rc, "bg_rollout_rows")
            if self._rowbuf is not None:
                self.observe()
            return self._stats
        ob = obs_buffers or self._obs
        stride = 1 if (obs_buffers is not None and obs_buffers.steps > 1) else 0
        if stride and obs_buffers.steps < steps:
            raise ValueError("obs_buffers has fewer rows than steps")
        if not stride and steps > 1:
            # reward / terminated / actions share the observation's row stride (bg_rollout: row = env + t * N only when obs_stride_steps != 0):
            # without per-step observation buffers every step writes ROW 0 of them.  A [steps, N] tensor here would come back with one filled row.
            for name, tns in (("reward", reward), ("terminated", terminated), ("actions", actions)):
                if tns is not None and tns.dim() > 1 and tns.shape[0] > 1:
                    raise ValueError(f"{name} has {tns.shape[0]} rows but no per-step obs_buffers were given: every step would overwrite row 0 "
                                     f"(pass ObsBuffers(n, device, steps={steps}), or a [N] tensor for the last step's values)")
        if zero_stats:
            self._stats.zero_()
        with torch.cuda.device(self.device):
            self._check(self._L.bg_rollout(
                self._h, int(steps), int(policy), C.c_uint64(policy_seed), C.c_uint64(env_index0), C.c_uint64(t0),
                C.byref(ob.ptrs), stride,
                None if reward is None else C.c_void_p(reward.data_ptr()),
                None if terminated is None else C.c_void_p(terminated.data_ptr()),
                None if actions is None else C.c_void_p(actions.data_ptr()),
                C.c_void_p(self._stats.data_ptr()), self._stream()), "bg_rollout")
        if self._rowbuf is not None:
            self.observe()   # (obs_layout "rows": the live records follow)
        return self._stats

    def stats(self) -> Dict[str, int]:
        """Aggregate counters of the rollouts since the last zeroing.  Also checks the device error word (a look-ahead ring
        that ran dry would otherwise go unnoticed behind a return code of 0)."""
        self.check()
        s = self._stats.cpu().numpy()
        u = s.view(np.uint64)
        return {"steps": int(u[0]), "episodes": int(u[1]), "plays": int(u[2]), "score_sum": int(s[3]),
                "reward_bits": int(u[4]), "obs_hash": int(u[5])}

    def inject(self, jokers=None, money=None, ante=None, levels=None, mask=None, apply_now: bool = True):
        """Harness injection (reset template + optionally the live state): jokers = list of id lists per env."""
        n = self.num_envs
        jptr = nptr = mptr = aptr = lptr = kptr = None
        keep = []
        if jokers is not None:
            ja = np.zeros((n, 5), np.int32)
            na = np.zeros(n, np.int32)
            for i, js in enumerate(jokers):
                na[i] = len(js)
                ja[i, :len(js)] = js
            keep += [ja, na]
            jptr, nptr = ja.ctypes.data_as(C.c_void_p), na.ctypes.data_as(C.c_void_p)
        if money is not None:
            ma = np.ascontiguousarray(np.asarray(money, np.int64)); keep.append(ma); mptr = ma.ctypes.data_as(C.c_void_p)
        if ante is not None:
            aa = np.ascontiguousarray(np.asarray(ante, np.int32)); keep.append(aa); aptr = aa.ctypes.data_as(C.c_void_p)
        if levels is not None:
            la = np.ascontiguousarray(np.asarray(levels, np.uint8)); keep.append(la); lptr = la.ctypes.data_as(C.c_void_p)
        if mask is not None:
            ka = np.ascontiguousarray(np.asarray(mask, np.uint8)); keep.append(ka); kptr = ka.ctypes.data_as(C.c_void_p)
        with torch.cuda.device(self.device):
            self._check(self._L.bg_inject(self._h, jptr, nptr, mptr, aptr, lptr, kptr, 1 if apply_now else 0,
                                          self._stream()), "bg_inject")

    def set_max_ante(self, max_ante, mask=None):
        """`CurriculumBalatroEnv.current_max_ante` (train_balatro_agent.py:129-166): one int for every (masked) env, or one
        per env.  0 = no cap.  Takes effect from the next step on; survives reset()."""
        mk = None if mask is None else np.ascontiguousarray(np.asarray(mask, np.uint8))
        per = None
        if not np.isscalar(max_ante):
            per = np.ascontiguousarray(np.asarray(max_ante, np.int32))
            if per.shape != (self.num_envs,):
                raise ValueError(f"per-env caps must have shape ({self.num_envs},)")
        with torch.cuda.device(self.device):
            self._check(self._L.bg_set_max_ante(
                self._h, 0 if per is not None else int(max_ante), None if per is None else per.ctypes.data_as(C.c_void_p),
                None if mk is None else mk.ctypes.data_as(C.c_void_p), self._stream()), "bg_set_max_ante")
        if per is None and mask is None:
            self.max_ante = int(max_ante)

    def inject_deck(self, decks, mask=None):
        """The live deck order per env ([N, 52] card codes (rank-2)*4+suit, each row a permutation): what a harness does by
        writing env.state.deck (balatro_env_2.py:528-531).  The next reset() reshuffles."""
        da = np.ascontiguousarray(np.asarray(decks, np.uint8))
        if da.shape != (self.num_envs, 52):
            raise ValueError(f"decks must have shape ({self.num_envs}, 52)")
        mk = None if mask is None else np.ascontiguousarray(np.asarray(mask, np.uint8))
        with torch.cuda.device(self.device):
            self._check(self._L.bg_inject_deck(self._h, da.ctypes.data_as(C.c_void_p),
                                               None if mk is None else mk.ctypes.data_as(C.c_void_p), self._stream()),
                        "bg_inject_deck")
        self.observe()

    def inject_cards(self, cards, mask=None, apply_now: bool = True):
        """Card states (cards.py CardState) per env: cards[i] = iterable of (deck_index, enhancement, edition, seal) codes.
        Re-applied after every reset, like `inject` (needs card_states=True)."""
        n = self.num_envs
        enh = np.zeros((n, 52), np.uint8); edi = np.zeros((n, 52), np.uint8); seal = np.zeros((n, 52), np.uint8)
        for i, cs in enumerate(cards):
            for (idx, e, d, s) in cs:
                enh[i, idx], edi[i, idx], seal[i, idx] = e, d, s
        mk = None if mask is None else np.ascontiguousarray(np.asarray(mask, np.uint8))
        with torch.cuda.device(self.device):
            self._check(self._L.bg_inject_cards(
                self._h, enh.ctypes.data_as(C.c_void_p), edi.ctypes.data_as(C.c_void_p), seal.ctypes.data_as(C.c_void_p),
                None if mk is None else mk.ctypes.data_as(C.c_void_p), 1 if apply_now else 0, self._stream()), "bg_inject_cards")
        if apply_now:
            self.observe()

    def inject_consumables(self, consumables, mask=None, apply_now: bool = True):
        """state.consumables per env (reset template, like `inject`): consumables[i] = up to 2 ids as in
        `_get_consumable_ids` (balatro_env_2.py:1545-1567): tarots 1-22, planets 30-41, spectrals 50-67.  Tarot and
        spectral cards edit card states, so they need card_states=True."""
        n = self.num_envs
        ids = np.zeros((n, 2), np.int32)
        cnt = np.zeros(n, np.int32)
        for i, cs in enumerate(consumables):
            cnt[i] = len(cs)
            ids[i, :len(cs)] = list(cs)[:2]
        mk = None if mask is None else np.ascontiguousarray(np.asarray(mask, np.uint8))
        with torch.cuda.device(self.device):
            self._check(self._L.bg_inject_consumables(
                self._h, ids.ctypes.data_as(C.c_void_p), cnt.ctypes.data_as(C.c_void_p),
                None if mk is None else mk.ctypes.data_as(C.c_void_p), 1 if apply_now else 0, self._stream()),
                "bg_inject_consumables")
        if apply_now:
            self.observe()

    def get_state(self, env_index: int) -> bytes:
        """save_state() (balatro_env_2.py:1575-1593) as a versioned binary blob."""
        nb = int(self._L.bg_state_blob_bytes(self._h))
        buf = (C.c_uint8 * nb)()
        self._check(self._L.bg_get_state(self._h, int(env_index), buf, nb), "bg_get_state")
        return bytes(buf)

    def set_state(self, env_index: int, blob: bytes):
        """load_state() (balatro_env_2.py:1595-1615)."""
        buf = (C.c_uint8 * len(blob)).from_buffer_copy(blob)
        self._check(self._L.bg_set_state(self._h, int(env_index), buf, len(blob)), "bg_set_state")

    @staticmethod
    def parse_state_blob(blob: bytes) -> Dict[str, object]:
        """Named views of a state blob (bg_get_state): what save_state() (balatro_env_2.py:1575-1593) leaves out -- the RNG streams --
        is in here, so tests can look at a full shuffled deck, the pre-shuffled decks, the raw global-stream blocks and the shop seeds."""
        b = np.frombuffer(blob, np.uint8)
        hdr = b[:16].view(np.uint32)
        kg, cards, ks, kd = int(hdr[2] & 0xffff), bool(hdr[2] & 0x10000), int(hdr[3] & 0xffff), int(hdr[3] >> 16)
        out: Dict[str, object] = {"magic": int(hdr[0]), "version": int(hdr[1]), "KG": kg, "KS": ks, "KD": kd, "card_states": cards}
        off = 16

        def take(name, nbytes, dtype=np.uint8, shape=None):
            nonlocal off
            a = b[off:off + nbytes].view(dtype)
            out[name] = a.reshape(shape) if shape else a
            off += nbytes
        take("hot", nat.BLOB_NHOT * 16, np.uint32, (nat.BLOB_NHOT, 4))
        take("deck", nat.BLOB_NDECK * 16)                       # 52 card codes (rank - 2) * 4 + suit, then padding
        take("cold", nat.BLOB_NCOLD * 16)
        take("tmpl", nat.BLOB_NTMPL * 16)
        take("ring_decks", kd * nat.BLOB_NDECK * 16, np.uint8, (kd, nat.BLOB_NDECK * 16))
        take("global_blocks", kg * nat.BLOB_MTS * 4, np.uint32, (kg, nat.BLOB_MTS))   # RAW MT19937 words (tempered on read)
        take("shop_slots", ks * nat.SHOP_SLOT_WORDS * 4, np.uint32, (ks, nat.SHOP_SLOT_WORDS))
        take("shop_overflow", nat.BLOB_MTS * 4, np.uint32)
        take("deck_stream", nat.BLOB_MTS * 4, np.uint32)
        take("shopgen_stream", nat.BLOB_MTS * 4, np.uint32)
        take("shop_seed_ring", nat.BLOB_SSEED * 4, np.uint32)
        take("shop_seed_meta", 4, np.uint32)
        take("producers", 4, np.uint32)
        if cards:
            take("card_states", nat.BLOB_NCST * 16, np.uint16)
            take("card_template", nat.BLOB_NCST * 16, np.uint16)
            take("card_stream", nat.BLOB_MTS * 4, np.uint32)
            take("seal_stream", nat.BLOB_MTS * 4, np.uint32)
        if off != len(b):
            raise ValueError(f"state blob of {len(b)} bytes does not parse ({off} bytes understood)")
        out["deck"] = out["deck"][:52]
        out["ring_decks"] = out["ring_decks"][:, :52]
        out["shop_slot_seeds"] = out["shop_slots"][:, nat.SHOP_SLOT_SEED_WORD]
        hot7 = out["hot"][7]
        out["shop_slot_current"] = int(hot7[1] & 0xff)
        # words of the per-env global stream consumed since it was seeded (the block counter is a byte: modulo 256 blocks of 624 words)
        out["global_words_consumed"] = int((out["hot"][6][3] >> 24) & 0xff) * 624 + int(hot7[0] & 0xffff)
        return out

    def set_profiling(self, enable: bool):
        self._check(self._L.bg_set_profiling(self._h, 1 if enable else 0), "bg_set_profiling")

    def get_profile(self) -> Dict[str, float]:
        """Per-kernel HIP-event timings since the last call (synchronises)."""
        out = (C.c_double * 8)()
        self._check(self._L.bg_get_profile(self._h, out), "bg_get_profile")
        return {"rollout_ms": out[0], "rollout_launches": int(out[1]), "rollout_fused_steps": int(out[2]),
                "refill_ms": out[3], "refill_launches": int(out[4]), "step_ms": out[5], "step_launches": int(out[6])}

    def check(self):
        """Raise if the device reported an internal invariant violation (RNG look-ahead underflow)."""
        with torch.cuda.device(self.device):
            self._check(self._L.bg_check(self._h, self._stream()), "bg_check")


# ---------------------------------------------------------------------- operator-level entry points
def classify_batch(cards: torch.Tensor, n: torch.Tensor, lanes_per_case: int = 1, timing: bool = False):
    """`BalatroGame._classify_hand` (balatro_game.py:40-93) for M hands at once: cards uint8 [M, 8] card codes
    (rank-2)*4+suit, n uint8 [M] valid cards per row -> uint8 [M] HandType values.  Device tensors in, device tensor out.
    lanes_per_case: 1 (lane = hand) or 8 (lane = card, shuffle reductions inside 8-lane groups); identical results.
    timing=True returns (out, kernel milliseconds)."""
    L = nat.load()
    if cards.dtype != torch.uint8 or n.dtype != torch.uint8 or cards.dim() != 2 or cards.shape[1] != 8 or not cards.is_cuda \
            or tuple(n.shape) != (cards.shape[0],) or n.device != cards.device:
        raise ValueError("cards must be a uint8 [M, 8] device tensor, n a uint8 [M] tensor on the same device")
    cards, n = cards.contiguous(), n.contiguous()
    out = torch.empty(cards.shape[0], dtype=torch.uint8, device=cards.device)
    ms = C.c_float(0.0)
    with torch.cuda.device(cards.device):
        rc = L.bg_classify_batch_ex(C.c_void_p(cards.data_ptr()), C.c_void_p(n.data_ptr()), C.c_void_p(out.data_ptr()),
                                    C.c_int64(cards.shape[0]), int(lanes_per_case), C.byref(ms) if timing else None,
                                    C.c_void_p(torch.cuda.current_stream(cards.device).cuda_stream))
    if rc != 0:
        raise nat.NativeError(f"bg_classify_batch failed ({rc}): {L.bg_last_error(None).decode()}")
    return (out, float(ms.value)) if timing else out


def score_hand_batch(cases: torch.Tensor, lanes_per_case: int = 1, timing: bool = False):
    """`UnifiedScorer.score_hand` (unified_scoring.py:111-299) with joker NAMES for M cases at once: cases int32
    [M, 40] (layout: include/balatro_mi355x.h bg_score_hand_batch) -> int64 [M, 8] (score, chips, mult, x_mult bits, money,
    global-stream words consumed, next getrandbits(32), 0).  lanes_per_case / timing as in classify_batch."""
    L = nat.load()
    if cases.dtype != torch.int32 or cases.dim() != 2 or cases.shape[1] != nat.SCORE_CASE_WORDS or not cases.is_cuda:
        raise ValueError(f"cases must be an int32 [M, {nat.SCORE_CASE_WORDS}] device tensor")
    cases = cases.contiguous()
    out = torch.zeros((cases.shape[0], nat.SCORE_OUT_WORDS), dtype=torch.int64, device=cases.device)
    ms = C.c_float(0.0)
    with torch.cuda.device(cases.device):
        rc = L.bg_score_hand_batch_ex(C.c_void_p(cases.data_ptr()), C.c_void_p(out.data_ptr()), int(cases.shape[0]),
                                      int(lanes_per_case), C.byref(ms) if timing else None,
                                      C.c_void_p(torch.cuda.current_stream(cases.device).cuda_stream))
    if rc != 0:
        raise nat.NativeError(f"bg_score_hand_batch failed ({rc}): {L.bg_last_error(None).decode()}")
    return (out, float(ms.value)) if timing else out


def sim_evaluate_batch(hands: torch.Tensor, n: torch.Tensor, flags: torch.Tensor) -> torch.Tensor:
    """`BalatroSimulator.evaluate_hand` (balatro_sim.py:220-366) for M hands: hands int32 [M, 8, 6] sim cards (rank, suit,
    base_value, enhancement, edition, seal), n int32 [M], flags int32 [M] (1 Four Fingers, 2 Shortcut) -> int8 [M, 128]
    (layout: include/balatro_mi355x.h bg_sim_evaluate_batch)."""
    L = nat.load()
    if hands.dtype != torch.int32 or hands.dim() != 3 or tuple(hands.shape[1:]) != (8, 6) or not hands.is_cuda:
        raise ValueError("hands must be an int32 [M, 8, 6] device tensor")
    if tuple(n.shape) != (hands.shape[0],) or tuple(flags.shape) != (hands.shape[0],):
        raise ValueError("n and flags must have shape [M]")
    hands = hands.contiguous()   # n / flags may come from the host or as other integer types: moved and converted, never handed over as they are
    n, flags = n.to(device=hands.device, dtype=torch.int32).contiguous(), flags.to(device=hands.device, dtype=torch.int32).contiguous()
    out = torch.zeros((hands.shape[0], nat.SIM_EVAL_BYTES), dtype=torch.int8, device=hands.device)
    with torch.cuda.device(hands.device):
        rc = L.bg_sim_evaluate_batch(C.c_void_p(hands.data_ptr()), C.c_void_p(n.data_ptr()), C.c_void_p(flags.data_ptr()),
                                     C.c_void_p(out.data_ptr()), int(hands.shape[0]),
                                     C.c_void_p(torch.cuda.current_stream(hands.device).cuda_stream))
    if rc != 0:
        raise nat.NativeError(f"bg_sim_evaluate_batch failed ({rc}): {L.bg_last_error(None).decode()}")
    return out


def sim_score_batch(cases: torch.Tensor) -> torch.Tensor:
    """`BalatroSimulator.calculate_score` (balatro_sim.py:402-548) for M cases: int32 [M, 64] -> int64 [M, 8]
    (layouts: include/balatro_mi355x.h bg_sim_score_batch)."""
    L = nat.load()
    if cases.dtype != torch.int32 or cases.dim() != 2 or cases.shape[1] != nat.SIM_CASE_WORDS or not cases.is_cuda:
        raise ValueError(f"cases must be an int32 [M, {nat.SIM_CASE_WORDS}] device tensor")
    cases = cases.contiguous()
    out = torch.zeros((cases.shape[0], 8), dtype=torch.int64, device=cases.device)
    with torch.cuda.device(cases.device):
        rc = L.bg_sim_score_batch(C.c_void_p(cases.data_ptr()), C.c_void_p(out.data_ptr()), int(cases.shape[0]),
                                  C.c_void_p(torch.cuda.current_stream(cases.device).cuda_stream))
    if rc != 0:
        raise nat.NativeError(f"bg_sim_score_batch failed ({rc}): {L.bg_last_error(None).decode()}")
    return out
