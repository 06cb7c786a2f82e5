#!/bin/bash
# the refill in pieces beside EVERY launch length (BG_REFILL_SLICED_DIV=1) against pieces beside short launches only (4, shipped): the default 372-step shape, 180, 100, 20
out=gpurun_out/r05ag; mkdir -p $out; export TMPDIR=/tmp; export BALATRO_MI355X_LIB=build/variants/div.so
for rep in 1 2 3; do for div in 4 1 2; do
  BG_REFILL_SLICED_DIV=$div timeout 300 python bench.py --no-cpu-baseline --no-step-path --no-small-n > $out/default_div${div}_$rep.json 2>/dev/null
  BG_REFILL_SLICED_DIV=$div timeout 300 python bench.py --no-cpu-baseline --no-step-path --no-small-n --steps 180 --warmup 5 --samples 40 > $out/T180_div${div}_$rep.json 2>/dev/null
  BG_REFILL_SLICED_DIV=$div timeout 300 python bench.py --no-cpu-baseline --no-step-path --no-small-n --steps 20 --warmup 5 > $out/T20_div${div}_$rep.json 2>/dev/null
done; done
for f in $out/*.json; do python -c "
import json; d=json.loads(open('$f').read().strip().splitlines()[-1]); r=d['roofline']; s=d['samples']; print('$f'.split('/')[-1], 'value', round(d['value']/1e9,3), 'kfrac', round(r['kernel_frac'],4), 'launch_us', round(r['mean_launch_us'],1), 'sust', round(d['sustained']['value']/1e9,3), 'median', round(s['median']/1e9,3), 'p10', round(s['p10']/1e9,3), 'min', round(s['min']/1e9,3), 'min/med', round(s['min_over_median'],3))"; done | tee $out/summary.txt
