"""The constant tables shared as DATA by the oracle (oracle/bo_tables.h) and the product (balatro_gym_amd/csrc/bg_tables.h) come from one
generator (tools/gen_tables.py), so a wrong entry would pass every HIP-vs-oracle comparison.  Pin every entry independently of the
generator: against numpy / CPython arithmetic evaluated HERE (what the Python reference computes: `3.0 * np.log10(max(1, final_score))`
balatro_env_2.py:821, `0.8 ** debuffed_cards` boss_blinds.py:436, `ANTE_COST_MULT ** (ante - 1)` shop.py:105, `1.5 ** (ante - 8)`
balatro_env_2.py:73) and, when the reference is present, against its own joker data (jokers.py:11-161)."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADERS = {"BO": os.path.join(ROOT, "oracle", "bo_tables.h"), "BG": os.path.join(ROOT, "balatro_gym_amd", "csrc", "bg_tables.h")}


def _tables(prefix):
    text = open(HEADERS[prefix]).read()
    out = {}
    for m in re.finditer(r"static const (double|unsigned char) %s_(\w+)\[[^\]]*\] = \{([^}]*)\}" % prefix, text):
        kind, name, body = m.groups()
        items = [x.strip() for x in body.replace("\n", " ").split(",") if x.strip()]
        out[name] = [float.fromhex(x) for x in items] if kind == "double" else [int(x) for x in items]
    return out


def test_both_headers_hold_the_same_data():
    a, b = _tables("BO"), _tables("BG")
    assert set(a) == set(b) and {"LOG10", "POW08", "POW115", "POW15", "JOKER_COST"} <= set(a)
    for k in a:
        assert a[k] == b[k], k


@pytest.mark.parametrize("prefix", ["BO", "BG"])
def test_every_entry_equals_what_the_reference_arithmetic_gives_here(prefix):
    t = _tables(prefix)
    assert len(t["LOG10"]) == 2155 and len(t["POW08"]) == 9 and len(t["POW115"]) == 101 and len(t["POW15"]) == 93
    want = [float(np.log10(max(1, s))) for s in range(2155)]
    assert t["LOG10"] == want                                   # bit-exact (hex float literals)
    assert 3.0 * t["LOG10"][2154] < 10.0 <= 3.0 * float(np.log10(2155))   # the table ends where min(10, .) takes over
    assert t["POW08"] == [0.8 ** n for n in range(9)]
    assert t["POW115"] == [1.15 ** k for k in range(101)]
    assert t["POW15"] == [1.5 ** k for k in range(93)]


def test_joker_costs_equal_the_references_joker_data():
    from oracle import refharness as rh
    if not rh.reference_available():
        pytest.skip("reference not present (GPU box)")
    ref = rh.load_reference()
    rj = ref["jokers"] if isinstance(ref, dict) and "jokers" in ref else __import__("balatro_gym.jokers", fromlist=["x"])
    cost = _tables("BG")["JOKER_COST"]
    assert len(cost) == 151 and cost[0] == 0
    seen = 0
    for info in rj.JOKER_LIBRARY:
        jid, base = int(info.id), int(info.base_cost)
        if 1 <= jid <= 150:
            seen += 1
            assert cost[jid] == base, (jid, info.name)
    assert seen == 150
