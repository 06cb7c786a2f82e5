#!/bin/bash
# round 5, GPU box: eager refill beside 20-step launches -- kernels one after the other (order 2) or side by side (0 / 1), every launch (min 16) or every other one (40)
out=gpurun_out/r05c; mkdir -p $out; export TMPDIR=/tmp
./tools/micro/recwrite > $out/recwrite.txt 2>&1; tail -5 $out/recwrite.txt
for rep in 1 2; do for cfg in "0 2" "16 0" "16 1" "16 2" "40 0" "40 2"; do set -- $cfg
  BG_REFILL_MIN=$1 BG_REFILL_ORDER=$2 timeout 200 python bench.py --no-cpu-baseline --no-step-path --no-small-n --steps 20 --warmup 5 > $out/T20_min$1_order$2_$rep.json 2>/dev/null
done; done
for f in $out/*.json; do python -c "
import json; d=json.loads(open('$f').read().strip().splitlines()[-1]); r=d['roofline']; print('$f'.split('/')[-1], 'value', round(d['value']/1e9,3), 'frac', round(r['frac'],4), 'kfrac', round(r['kernel_frac'],4), 'sust', round(d['sustained']['value']/1e9,3), 'launch_us', round(r['mean_launch_us'],1), 'median', round(d['samples']['median']/1e9,3), 'min', round(d['samples']['min']/1e9,3), 'refill_us', round(r['refill_kernel_us_in_timed_region'],1))"; done
