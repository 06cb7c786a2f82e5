#!/usr/bin/env python3
"""Aggregate a rocprofv3 PC-sampling CSV: samples per source line (Instruction_Comment) and per instruction, for kernels matching a filter."""
import collections, csv, glob, sys
path = sys.argv[1]
files = glob.glob(path + "/**/*pc_sampling*.csv", recursive=True)
print("files:", files)
by_line, by_ins, total = collections.Counter(), collections.Counter(), 0
cols = None
for f in files:
    with open(f) as fh:
        rd = csv.DictReader(fh)
        cols = rd.fieldnames
        for r in rd:
            total += 1
            by_line[r.get("Instruction_Comment", "")] += 1
            by_ins[(r.get("Instruction_Comment", ""), r.get("Instruction", ""))] += 1
print("columns:", cols, "samples:", total)
print("---- top source lines")
for k, v in by_line.most_common(80):
    print(f"{v:8d} {100.0 * v / max(total, 1):5.1f}%  {k}")
print("---- top instructions")
for (c, i), v in by_ins.most_common(80):
    print(f"{v:8d} {100.0 * v / max(total, 1):5.1f}%  {i:60s} {c}")
