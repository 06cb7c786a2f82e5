#!/bin/bash
# A/B of the step engine's queue thresholds and service-wave masks on one box
cd "$(dirname "$0")/.."
python tools/bench_brief.py > /dev/null
python tools/bench_brief.py
for part in 1 16 32; do for p in 32 48 64; do
  BG_ENG_PART=$part BG_ENG_PLAY=$p BG_ENG_OTHER=$p python tools/bench_brief.py
done; done
for m in 0x78 0x7c 0x7e 0x7f 0x3c; do BG_ENG_SMASK=$(($m)) python tools/bench_brief.py; done
python tools/bench_brief.py
