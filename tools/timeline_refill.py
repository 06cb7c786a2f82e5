#!/usr/bin/env python3
"""Development: from a rocprofv3 --kernel-trace CSV, the engine launches of a run of 20-step launches and which refill kernels ran beside each
(begin / end in us relative to the first engine launch listed).  usage: timeline_refill.py <kernel_trace.csv> [first] [count]"""
import csv, sys
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].split("<")[0]))
rows.sort()
eng = [i for i, r in enumerate(rows) if "engine3" in r[2]]
durs = [(rows[i][1] - rows[i][0]) / 1e3 for i in eng]
import statistics
med = statistics.median(durs)
print(f"{len(eng)} engine launches, median {med:.1f} us, max {max(durs):.1f} us")
slow = [k for k, d in enumerate(durs) if d > 1.5 * med]
print("slow launches (index: us):", [(k, round(durs[k], 1)) for k in slow][:40])
first = int(sys.argv[2]) if len(sys.argv) > 2 else (max(0, slow[len(slow) // 2] - 2) if slow else 0)
count = int(sys.argv[3]) if len(sys.argv) > 3 else 14
t0 = rows[eng[first]][0]
lo, hi = rows[eng[first]][0], rows[eng[min(first + count, len(eng) - 1)]][1]
for s, e, n in rows:
    if e >= lo and s <= hi:
        print(f"  {(s - t0) / 1e3:9.1f} .. {(e - t0) / 1e3:9.1f}  ({(e - s) / 1e3:7.1f} us)  {n}")
