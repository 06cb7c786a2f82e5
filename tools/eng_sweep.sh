#!/bin/bash
# A/B of the step engine's queue thresholds on one box
cd "$(dirname "$0")/.."
python tools/bench_brief.py
for part in 16 32 48 255; do for p in 32 48 64; do
  BG_ENG_PART=$part BG_ENG_PLAY=$p BG_ENG_OTHER=$p python tools/bench_brief.py
done; done
