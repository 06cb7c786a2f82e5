#!/bin/bash
# development helper (GPU box): bench at several fused-chunk lengths; usage: tools/sweep.sh <tag> "<chunk KG KS KD>" ...
tag=$1; shift
mkdir -p gpurun_out/sw
for cfg in "$@"; do
  set -- $cfg
  BG_BENCH_CHUNK=$1 BG_KG=$2 BG_KS=$3 BG_KD=$4 python bench.py --steps 1536 --warmup 192 --no-cpu-baseline > gpurun_out/sw/${tag}_c$1.json 2> gpurun_out/sw/${tag}_c$1.err
  python - <<P
import json
d=json.load(open("gpurun_out/sw/${tag}_c$1.json")); r=d["roofline"]
print("${tag}", $1, "Msteps/s", round(d["value"]/1e6,1), "rollout_us", round(r["mean_launch_us"],1), "refill_us", round(r["refill_mean_launch_us"],1))
P
done
