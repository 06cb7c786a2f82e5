#!/bin/bash
# A/B of the step engine's queue thresholds (and the service-wave kernel for reference) on one box
cd "$(dirname "$0")/.."
python tools/bench_brief.py
BG_ROLLOUT_V=3 python tools/bench_brief.py
for part in 16 32 48 255; do
  BG_ENG_PART=$part python tools/bench_brief.py
done
for part in 32 255; do for p in 32 48; do
  BG_ENG_PART=$part BG_ENG_PLAY=$p BG_ENG_OTHER=$p python tools/bench_brief.py
done; done
BG_ENG_PART=255 python tools/bench_brief.py --steps 20 --warmup 5
BG_ENG_PART=255 BALATRO_MI355X_LIB=balatro_gym_amd/variants/t4.so python tools/timing4.py
BG_ENG_PART=32 BALATRO_MI355X_LIB=balatro_gym_amd/variants/t4.so python tools/timing4.py
