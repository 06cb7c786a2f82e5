#!/bin/bash
# Development aid: worker / copier wave counts of a 20-step launch (the driver's shape), two rounds
cd "$(dirname "$0")/.." || exit 1
for r in 1 2; do
for w in 3 4 5; do for c in 1 2 3; do
  BG_ENG_WAVES=$w BG_ENG_COPIERS=$c timeout 120 python tools/bench_brief.py --samples 2 --steps 20 --warmup 5
done; done; done
