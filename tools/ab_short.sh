#!/bin/bash
# GPU box: the driver's launch shape (--steps 20 --warmup 5) under different service thresholds
cd "$(dirname "$0")/.."
for r in 1 2; do
for cfg in "" "BG_ENG_PLAY=1 BG_ENG_OTHER=1" "BG_ENG_PLAY=8 BG_ENG_OTHER=8" "BG_ENG_PLAY=16 BG_ENG_OTHER=16" "BG_ENG_PLAY=32 BG_ENG_OTHER=32"; do
  env $cfg python tools/bench_brief.py --steps 20 --warmup 5 | sed "s/^/[$cfg] /"
done; done
