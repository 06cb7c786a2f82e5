#!/usr/bin/env python3
"""Dump the per-kernel summary (rocprofv3 --kernel-trace --stats, rocpd SQLite output) as text for profiles/."""
import sqlite3
import sys


def main(db_path, out_path=None):
    db = sqlite3.connect(db_path)
    rows = list(db.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    lines = [f"# rocprofv3 --kernel-trace --stats summary of {db_path} (durations in microseconds)",
             f"{'calls':>8} {'total_us':>14} {'avg_us':>12} {'pct':>7}  kernel"]
    for name, calls, total, avg, pct in rows:
        short = name.split("(")[0]
        lines.append(f"{calls:>8} {total:>14.1f} {avg:>12.2f} {pct:>7.2f}  {short}")
    text = "\n".join(lines) + "\n"
    if out_path:
        open(out_path, "w").write(text)
    print(text)


if __name__ == "__main__":
    main(*sys.argv[1:3])
