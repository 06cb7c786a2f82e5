#!/bin/bash
# development helper (GPU box): phase-B thresholds "play other ready" for the service-wave kernel (BG_ROLLOUT_V=3)
run() { set -- $1; BG_ROLLOUT_V=${V:-3} BG_TH_PLAY=$1 BG_TH_OTHER=$2 BG_TH_READY=$3 python bench.py --no-cpu-baseline --warmup 7440 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('v${V:-3} th $1 $2 $3', round(d['value']/1e9,3), 'G rollout_us', round(d['roofline']['mean_launch_us'],1))"; }
for th in "$@"; do run "$th"; done
