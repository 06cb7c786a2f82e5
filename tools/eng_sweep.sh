#!/bin/bash
# A/B of the step engine's queue thresholds on one box (interleaved with the default)
cd "$(dirname "$0")/.."
python tools/bench_brief.py > /dev/null
for cfg in "" "BG_ENG_PLAY=32 BG_ENG_OTHER=32" "" "BG_ENG_PLAY=48 BG_ENG_OTHER=48" "" "BG_ENG_PART=16" "" "BG_ENG_MORE=1" "" "BG_ENG_MORE=2" "" "BG_ENG_SMASK=56" ""; do
  env $cfg python tools/bench_brief.py
done
