"""ctypes binding of the CPU oracle (TEST INFRASTRUCTURE; see oracle/balatro_oracle.h).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# BALATRO_ORACLE_LIB: another build of the same sources -- the AddressSanitizer / UBSan build (`make -C oracle asan`) that
# tests/test_oracle_sanitized.py runs the golden tests against
LIB_PATH = os.environ.get("BALATRO_ORACLE_LIB") or os.path.join(HERE, "libbalatro_oracle.so")

NACT = 60
FLAG_SCORER_JOKERS = 1
NAMES_ENV, NAMES_SIM = 0, 1
POLICY_UNIFORM, POLICY_SMALL_ONLY, POLICY_CYCLE3 = 0, 1, 2

OBS_KEYS = [
    "hand", "hand_size", "deck_size", "selected_cards", "chips_scored", "round_chips_scored", "progress_ratio",
    "mult", "chips_needed", "money", "ante", "round", "hands_left", "discards_left", "joker_count", "joker_ids",
    "joker_slots", "consumable_count", "consumables", "consumable_slots", "shop_items", "shop_costs",
    "shop_rerolls", "hand_levels", "phase", "action_mask", "hands_played", "best_hand_this_ante",
    "boss_blind_active", "boss_blind_type", "face_down_cards",
]


class Obs(C.Structure):
    _fields_ = [
        ("selected_cards", C.c_int64 * 8), ("face_down_cards", C.c_int64 * 8), ("chips_scored", C.c_int64),
        ("round_chips_scored", C.c_int32), ("progress_ratio", C.c_float), ("mult", C.c_int32),
        ("chips_needed", C.c_int32), ("money", C.c_int32), ("hands_played", C.c_int32),
        ("best_hand_this_ante", C.c_int32), ("ante", C.c_int16), ("joker_ids", C.c_int16 * 10),
        ("consumables", C.c_int16 * 5), ("shop_items", C.c_int16 * 10), ("shop_costs", C.c_int16 * 10),
        ("shop_rerolls", C.c_int16), ("hand", C.c_int8 * 8), ("hand_size", C.c_int8), ("deck_size", C.c_int8),
        ("round", C.c_int8), ("hands_left", C.c_int8), ("discards_left", C.c_int8), ("joker_count", C.c_int8),
        ("joker_slots", C.c_int8), ("consumable_count", C.c_int8), ("consumable_slots", C.c_int8),
        ("hand_levels", C.c_int8 * 12), ("phase", C.c_int8), ("action_mask", C.c_int8 * NACT),
        ("boss_blind_active", C.c_int8), ("boss_blind_type", C.c_int8),
    ]


# reference dtypes (balatro_env_2.py:1488-1531)
OBS_DTYPES = {
    "hand": np.int8, "hand_size": np.int8, "deck_size": np.int8, "selected_cards": np.int64,
    "chips_scored": np.int64, "round_chips_scored": np.int32, "progress_ratio": np.float32, "mult": np.int32,
    "chips_needed": np.int32, "money": np.int32, "ante": np.int16, "round": np.int8, "hands_left": np.int8,
    "discards_left": np.int8, "joker_count": np.int8, "joker_ids": np.int16, "joker_slots": np.int8,
    "consumable_count": np.int8, "consumables": np.int16, "consumable_slots": np.int8, "shop_items": np.int16,
    "shop_costs": np.int16, "shop_rerolls": np.int16, "hand_levels": np.int8, "phase": np.int8,
    "action_mask": np.int8, "hands_played": np.int32, "best_hand_this_ante": np.int32,
    "boss_blind_active": np.int8, "boss_blind_type": np.int8, "face_down_cards": np.int64,
}
OBS_SHAPES = {
    "hand": (8,), "selected_cards": (8,), "joker_ids": (10,), "consumables": (5,), "shop_items": (10,),
    "shop_costs": (10,), "hand_levels": (12,), "action_mask": (NACT,), "face_down_cards": (8,),
}


class Info(C.Structure):
    _fields_ = [
        ("final_score", C.c_int64), ("reward_terms", C.c_double * 8), ("error", C.c_int32), ("flags", C.c_int32),
        ("aux", C.c_int32), ("hand_type", C.c_int8), ("cards_played", C.c_int8), ("breakdown", C.c_double * 8),
    ]


class SCard(C.Structure):
    _fields_ = [("rank", C.c_int32), ("suit", C.c_int32), ("chips", C.c_int32)]


class ScoreOut(C.Structure):
    _fields_ = [("score", C.c_int64), ("chips", C.c_int64), ("mult", C.c_int64), ("x_mult", C.c_double),
                ("money", C.c_int64), ("draws", C.c_int32)]


class SimCard(C.Structure):
    _fields_ = [("rank", C.c_int32), ("suit", C.c_int32), ("base_value", C.c_int32), ("enhancement", C.c_int32),
                ("edition", C.c_int32), ("seal", C.c_int32)]


class SimEval(C.Structure):
    _fields_ = [("top", C.c_int8), ("nlists", C.c_int8 * 12), ("n0", C.c_int8 * 12), ("pos", (C.c_int8 * 8) * 12)]


class SimScoreOut(C.Structure):
    _fields_ = [("score", C.c_int64), ("chips", C.c_int64), ("add_mult", C.c_int64), ("x_mult", C.c_double),
                ("money", C.c_int64), ("draws", C.c_int32), ("top", C.c_int32), ("nscoring", C.c_int32)]


class MT(C.Structure):
    _fields_ = [("mt", C.c_uint32 * 624), ("mti", C.c_int32)]


_lib = None


def build(force: bool = False) -> str:
    """Compile oracle/libbalatro_oracle.so with gcc (building the checker is not using it)."""
    src = os.path.join(HERE, "balatro_oracle.c")
    deps = [src, os.path.join(HERE, "bo_sim.c"), os.path.join(HERE, "balatro_oracle.h"), os.path.join(HERE, "bo_tables.h")]
    if os.environ.get("BALATRO_ORACLE_LIB"):
        return LIB_PATH
    if force or not os.path.exists(LIB_PATH) or any(os.path.getmtime(d) > os.path.getmtime(LIB_PATH) for d in deps):
        subprocess.check_call(["make", "-C", HERE, "-s", "libbalatro_oracle.so"])
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB_PATH)
        L.bo_create.restype = C.c_void_p
        L.bo_create.argtypes = [C.c_uint32, C.c_int32]
        L.bo_destroy.argtypes = [C.c_void_p]
        L.bo_construct.argtypes = [C.c_void_p, C.c_int64]
        L.bo_reset.argtypes = [C.c_void_p, C.c_int, C.c_int64]
        L.bo_get_obs.argtypes = [C.c_void_p, C.POINTER(Obs)]
        L.bo_step.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_uint8), C.POINTER(Info)]
        L.bo_set_jokers.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.c_int]
        L.bo_set_card_state.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
        L.bo_set_hand_level.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.bo_set_template_jokers.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.c_int]
        L.bo_set_money.argtypes = [C.c_void_p, C.c_int64]
        L.bo_set_deck.argtypes = [C.c_void_p, C.POINTER(C.c_uint8)]
        L.bo_set_max_ante.argtypes = [C.c_void_p, C.c_int]
        L.bo_set_consumables.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.c_int]
        L.bo_set_ante.argtypes = [C.c_void_p, C.c_int]
        L.bo_policy_action.restype = C.c_int
        L.bo_policy_action.argtypes = [C.c_void_p, C.c_int, C.c_uint64, C.c_uint64, C.c_uint64]
        L.bo_classify.restype = C.c_int
        L.bo_classify.argtypes = [C.POINTER(C.c_uint8), C.c_int]
        L.bo_mt_seed.argtypes = [C.POINTER(MT), C.c_uint64]
        L.bo_mt_u32.restype = C.c_uint32
        L.bo_mt_u32.argtypes = [C.POINTER(MT)]
        L.bo_mt_randbelow.restype = C.c_uint32
        L.bo_mt_randbelow.argtypes = [C.POINTER(MT), C.c_uint32]
        L.bo_mt_random.restype = C.c_double
        L.bo_mt_random.argtypes = [C.POINTER(MT)]
        L.bo_mt_shuffle_u8.argtypes = [C.POINTER(MT), C.POINTER(C.c_uint8), C.c_int]
        L.bo_global_seed.restype = C.c_uint64
        L.bo_global_seed.argtypes = [C.c_int64]
        L.bo_policy_hash.restype = C.c_uint32
        L.bo_policy_hash.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64]
        L.bo_blind_chips.restype = C.c_int64
        L.bo_blind_chips.argtypes = [C.c_int, C.c_int]
        L.bo_score_hand.argtypes = [C.POINTER(SCard), C.c_int, C.POINTER(SCard), C.c_int, C.c_int, C.c_int, C.c_int,
                                    C.POINTER(C.c_int32), C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(MT),
                                    C.POINTER(ScoreOut)]
        L.bo_sim_evaluate.argtypes = [C.POINTER(SimCard), C.c_int, C.c_int, C.c_int, C.POINTER(SimEval)]
        L.bo_sim_score.argtypes = [C.POINTER(SimCard), C.c_int, C.POINTER(C.c_int32), C.c_int, C.c_int, C.c_int, C.c_int,
                                   C.POINTER(MT), C.POINTER(SimScoreOut)]
        L.bo_rollout.restype = C.c_int64
        L.bo_rollout.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int64, C.c_int, C.c_int, C.c_uint64, C.c_uint64,
                                 C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        _lib = L
    return _lib


def obs_to_dict(o: Obs) -> dict:
    d = {}
    for k in OBS_KEYS:
        v = getattr(o, k)
        if k in OBS_SHAPES:
            d[k] = np.array(list(v), dtype=OBS_DTYPES[k])
        else:
            d[k] = OBS_DTYPES[k](v)
    return d


class OracleEnv:
    """Single env with the reference's reset()/step() surface, backed by the C oracle."""

    def __init__(self, seed: int, scorer_jokers: bool = False, max_ante: int = 0):
        L = lib()
        self._L = L
        self._h = L.bo_create(FLAG_SCORER_JOKERS if scorer_jokers else 0, max_ante)
        L.bo_construct(self._h, seed)

    def __del__(self):
        try:
            if self._h:
                self._L.bo_destroy(self._h)
                self._h = None
        except Exception:
            pass

    @property
    def handle(self):
        return self._h

    def reset(self, seed=None):
        self._L.bo_reset(self._h, 0 if seed is None else 1, 0 if seed is None else seed)
        return self.obs()

    def obs(self):
        o = Obs()
        self._L.bo_get_obs(self._h, C.byref(o))
        return obs_to_dict(o)

    def step(self, action: int):
        r = C.c_double()
        t = C.c_uint8()
        info = Info()
        self._L.bo_step(self._h, int(action), C.byref(r), C.byref(t), C.byref(info))
        return self.obs(), r.value, bool(t.value), False, info

    def set_jokers(self, ids):
        arr = (C.c_int32 * len(ids))(*ids)
        self._L.bo_set_jokers(self._h, arr, len(ids))

    def set_card_state(self, deck_idx, enh=0, edi=0, seal=0):
        self._L.bo_set_card_state(self._h, deck_idx, enh, edi, seal)

    def set_hand_level(self, hand_type, level):
        self._L.bo_set_hand_level(self._h, hand_type, level)

    def set_template_jokers(self, ids):
        arr = (C.c_int32 * len(ids))(*ids)
        self._L.bo_set_template_jokers(self._h, arr, len(ids))

    def set_consumables(self, ids):
        arr = (C.c_int32 * max(1, len(ids)))(*ids)
        self._L.bo_set_consumables(self._h, arr, len(ids))

    def set_deck(self, codes):
        arr = (C.c_uint8 * 52)(*codes)
        self._L.bo_set_deck(self._h, arr)

    def set_max_ante(self, max_ante):
        self._L.bo_set_max_ante(self._h, int(max_ante))

    def set_money(self, money):
        self._L.bo_set_money(self._h, money)

    def set_ante(self, ante):
        self._L.bo_set_ante(self._h, ante)

    def policy_action(self, policy, policy_seed, env_index, t):
        return self._L.bo_policy_action(self._h, policy, policy_seed, env_index, t)


def classify(codes) -> int:
    arr = (C.c_uint8 * max(1, len(codes)))(*codes)
    return lib().bo_classify(arr, len(codes))


def score_hand(cards, scoring, hand_type, name_style, level, jokers, hands_left, discards_left, deck_len, seed):
    """cards/scoring: lists of (rank, suit, chips).  Returns (ScoreOut, draws) with a global stream seeded `seed`."""
    L = lib()
    ca = (SCard * max(1, len(cards)))(*[SCard(*c) for c in cards])
    sa = (SCard * max(1, len(scoring)))(*[SCard(*c) for c in scoring])
    ja = (C.c_int32 * max(1, len(jokers)))(*jokers)
    mt = MT()
    L.bo_mt_seed(C.byref(mt), seed)
    out = ScoreOut()
    L.bo_score_hand(ca, len(cards), sa, len(scoring), hand_type, name_style, level, ja, len(jokers), hands_left,
                    discards_left, deck_len, C.byref(mt), C.byref(out))
    return out


def sim_evaluate(cards, four_fingers=False, shortcut=False):
    """balatro_sim.py evaluate_hand: cards = [(rank, suit, base_value, enhancement, edition, seal)] -> (top, {type: (nlists, [positions of list 0])})."""
    L = lib()
    ca = (SimCard * max(1, len(cards)))(*[SimCard(*c) for c in cards])
    ev = SimEval()
    L.bo_sim_evaluate(ca, len(cards), int(four_fingers), int(shortcut), C.byref(ev))
    return int(ev.top), {t: (int(ev.nlists[t]), [int(ev.pos[t][i]) for i in range(ev.n0[t])]) for t in range(12)}


def sim_score(cards, jokers, hands_left, discards_left, deck_len, seed):
    """balatro_sim.py calculate_score after random.seed(seed); Four Fingers (18) / Shortcut (69) in `jokers` act on the evaluation."""
    L = lib()
    ca = (SimCard * max(1, len(cards)))(*[SimCard(*c) for c in cards])
    ja = (C.c_int32 * max(1, len(jokers)))(*jokers)
    mt = MT()
    L.bo_mt_seed(C.byref(mt), seed)
    out = SimScoreOut()
    L.bo_sim_score(ca, len(cards), ja, len(jokers), hands_left, discards_left, deck_len, C.byref(mt), C.byref(out))
    return out
