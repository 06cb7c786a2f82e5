#!/usr/bin/env python3
"""Mean per-launch value of every counter collected by tools/sq_counters.sh for the rollout kernel."""
import csv, glob, sys
tag = sys.argv[1]
kernel = sys.argv[2] if len(sys.argv) > 2 else "bg_engine_kernel"
for path in sorted(glob.glob(f"gpurun_out/{tag}/sq_*/runc_counter_collection.csv")):
    acc = {}
    for r in csv.DictReader(open(path)):
        if kernel in r["Kernel_Name"]:
            acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    for k, v in acc.items():
        v = v[len(v) // 4:]
        print(f"{k:32s} {sum(v) / len(v):18,.0f}   ({len(v)} launches)")
