#!/bin/bash
# mid-size launches (48 / 100 / 180 steps): the refill whole (BG_REFILL_SLICED=0), in pieces up to max_chunk / 4 (48 only), in pieces up to max_chunk / 2 and / 1
out=gpurun_out/r05af; mkdir -p $out; export TMPDIR=/tmp; export BALATRO_MI355X_LIB=build/variants/div.so
for rep in 1 2; do for T in 48 100 180; do for v in "0 4" "1 4" "1 2" "1 1"; do set -- $v
  BG_REFILL_SLICED=$1 BG_REFILL_SLICED_DIV=$2 timeout 300 python bench.py --no-cpu-baseline --no-step-path --no-small-n --steps $T --warmup 5 --samples 40 > $out/T${T}_sliced$1_div$2_$rep.json 2>/dev/null
done; done; done
for f in $out/*.json; do python -c "
import json; d=json.loads(open('$f').read().strip().splitlines()[-1]); r=d['roofline']; s=d['samples']; print('$f'.split('/')[-1], 'value', round(d['value']/1e9,3), 'kfrac', round(r['kernel_frac'],4), 'sust', round(d['sustained']['value']/1e9,3), 'median', round(s['median']/1e9,3), 'p10', round(s['p10']/1e9,3), 'min', round(s['min']/1e9,3), 'min/med', round(s['min_over_median'],3))"; done | tee $out/summary.txt
