#!/usr/bin/env python3
"""A/B of the two lane mappings of the operator-level kernels on the GPU ("benchmark both"): lane = case against
8 lanes per case (lane = card / joker, `__shfl_xor` reductions inside 8-lane groups).  Kernel-only times (HIP events inside the
library).  Prints one JSON line; `--out FILE` also writes it."""
import argparse, json, os, random, sys
from itertools import combinations
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from balatro_gym_amd import classify_batch, score_hand_batch

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out")
    ap.add_argument("--score-cases", type=int, default=1 << 20)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    combos = np.fromiter((c for combo in combinations(range(52), 5) for c in combo), dtype=np.uint8, count=2598960 * 5).reshape(-1, 5)
    cards = np.zeros((combos.shape[0], 8), np.uint8); cards[:, :5] = combos
    cards_d = torch.from_numpy(cards).to(dev); n_d = torch.full((cards.shape[0],), 5, dtype=torch.uint8, device=dev)
    res = {"classify": {"hands": int(cards.shape[0])}, "score_hand": {"cases": a.score_cases}}
    ref = None
    for lanes in (1, 8):
        ts = []
        for _ in range(7):
            out, ms = classify_batch(cards_d, n_d, lanes_per_case=lanes, timing=True); ts.append(ms)
        ref = out if ref is None else ref
        assert torch.equal(out, ref)
        res["classify"][f"lanes{lanes}_us"] = round(1e3 * float(np.median(ts)), 1)
    r = np.random.default_rng(7)
    M = a.score_cases
    rec = np.zeros((M, 40), np.int32)
    ncards = r.integers(1, 9, M)
    for k in range(8):
        rank = r.integers(2, 15, M); suit = r.integers(0, 4, M)
        rec[:, 3 * k] = rank; rec[:, 3 * k + 1] = suit; rec[:, 3 * k + 2] = np.where(rank == 14, 11, np.minimum(rank, 10))
    rec[:, 24] = ncards; rec[:, 25] = ncards; rec[:, 26] = r.integers(0, 9, M); rec[:, 27] = 0; rec[:, 28] = r.integers(1, 4, M)
    pool = np.array([1, 136, 27, 38, 61, 16, 34, 108, 23, 22, 53, 97, 50, 2, 3, 4, 5, 8, 9, 10, 13, 14, 15, 134, 135, 48, 128, 122, 72, 140,
                     31, 39, 40, 41, 101, 124, 26, 33, 104, 147, 118, 119, 116, 117], np.int32)
    rec[:, 29] = 5
    for j in range(5):
        rec[:, 30 + j] = pool[r.integers(0, len(pool), M)]
    rec[:, 35] = r.integers(1, 5, M); rec[:, 36] = r.integers(0, 4, M); rec[:, 37] = 44
    rec[:, 38] = r.integers(0, 2 ** 31, M).astype(np.int32)
    rec_d = torch.from_numpy(rec).to(dev)
    ref = None
    for lanes in (1, 8):
        ts = []
        for _ in range(5):
            out, ms = score_hand_batch(rec_d, lanes_per_case=lanes, timing=True); ts.append(ms)
        ref = out if ref is None else ref
        assert torch.equal(out, ref)
        res["score_hand"][f"lanes{lanes}_us"] = round(1e3 * float(np.median(ts)), 1)
    # one wave's worth: 8 cases -- the dependent chain of one play, which is what a latency-bound step engine would feel
    lat = {}
    for lanes in (1, 8):
        ts = []
        for _ in range(21):
            _, ms = score_hand_batch(rec_d[:8], lanes_per_case=lanes, timing=True); ts.append(ms)
        lat[f"lanes{lanes}_us"] = round(1e3 * float(np.median(ts)), 2)
    res["score_hand_8_cases_one_wave"] = lat
    for k in ("classify", "score_hand"):
        res[k]["l8_over_l1"] = round(res[k]["lanes8_us"] / res[k]["lanes1_us"], 2)
    line = json.dumps(res)
    print(line)
    if a.out:
        open(a.out, "w").write(line + "\n")

if __name__ == "__main__":
    main()
