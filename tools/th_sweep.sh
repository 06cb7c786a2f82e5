#!/bin/bash
# development helper (GPU box): synchronous-refill rollout time for several phase-B thresholds "play other ready"
for th in "$@"; do
  set -- $th
  BG_ASYNC_REFILL=0 BG_TH_PLAY=$1 BG_TH_OTHER=$2 BG_TH_READY=$3 python bench.py --no-cpu-baseline --steps 512 --warmup 128 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('th $1 $2 $3', 'rollout_us', round(d['roofline']['mean_launch_us'],1))"
done
