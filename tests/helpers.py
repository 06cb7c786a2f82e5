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

TRACES = ["c1_small_only", "c2_cycle3", "c3_jokers", "c5_uniform", "c5_uniform_rich", "cards_levels", "consumables",
          "consumables_scorer"]


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
