#!/usr/bin/env python3
"""Per-kernel call count / median / last-quartile mean duration from a rocprofv3 rocpd database (development aid)."""
import collections, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
d = collections.defaultdict(list)
for n, s, e in db.execute("select name,start,end from kernels order by start"):
    d[n.split("(")[0].replace("void ", "")].append((e - s) / 1e3)
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    if "bg_" not in k: continue
    tail = v[len(v) // 2:]
    print(f"{k:34s} calls {len(v):4d}  median {sorted(v)[len(v)//2]:9.1f} us  steady mean {sum(tail)/len(tail):9.1f} us  total {sum(v)/1e3:8.2f} ms")
