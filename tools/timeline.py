#!/usr/bin/env python3
"""Development aid: print the kernel timeline (start offset, duration, stream/queue) of a rocprofv3 rocpd database for the last
few engine launches -- shows how the refill kernels overlap the step engine."""
import sqlite3, sys
def main(path, last=3):
    db = sqlite3.connect(path)
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
    kd = [t for t in tabs if 'kernel_dispatch' in t][0]
    ks = [t for t in tabs if 'kernel_symbol' in t][0]
    cols = [r[1] for r in db.execute(f"pragma table_info({kd})")]
    q = 'queue_id' if 'queue_id' in cols else 'stream_id'
    rows = list(db.execute(f"select s.kernel_name, d.start, d.end, d.{q} from {kd} d join {ks} s on d.kernel_id=s.id order by d.start"))
    eng = [i for i, r in enumerate(rows) if 'bg_engine' in r[0]]
    i0 = eng[-int(last) - 1]
    t0 = rows[i0][1]
    for n, s, e, qid in rows[i0:]:
        n = n.split('(')[0].replace('void ', '')
        print(f"{(s - t0) / 1000:10.1f} {(e - t0) / 1000:10.1f} {(e - s) / 1000:9.1f}  q{qid}  {n[:60]}")
if __name__ == "__main__":
    main(*sys.argv[1:])
