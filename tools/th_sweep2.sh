#!/bin/bash
# development helper (GPU box): default bench (async refill, 372-step launches) for several phase-B thresholds "play other ready";
# the baseline triple is repeated between candidates so that drift on the box shows up
base="40 40 24"
run() { set -- $1; BG_TH_PLAY=$1 BG_TH_OTHER=$2 BG_TH_READY=$3 python bench.py --no-cpu-baseline --warmup 7440 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('th $1 $2 $3', round(d['value']/1e9,3), 'G rollout_us', round(d['roofline']['mean_launch_us'],1))"; }
run "$base"; run "$base"
for th in "$@"; do run "$th"; run "$base"; done
