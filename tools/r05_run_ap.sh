#!/bin/bash
# the last launch of a refill period left free of pieces (16 pieces: the deck's last pass is one piece) against the library before it, interleaved on one box
out=gpurun_out/r05aq; mkdir -p $out; export TMPDIR=/tmp
for rep in 1 2 3; do for v in prev new; do
  lib=build/variants/prev.so; [ $v = new ] && lib=balatro_gym_amd/libbalatro_mi355x.so
  BALATRO_MI355X_LIB=$lib timeout 300 python bench.py --no-cpu-baseline --no-step-path --no-small-n --steps 20 --warmup 5 --samples 120 > $out/T20_${v}_$rep.json 2>/dev/null
  for T in 100 180; do BALATRO_MI355X_LIB=$lib timeout 300 python bench.py --no-cpu-baseline --no-step-path --no-small-n --steps $T --warmup 5 --samples 40 > $out/T${T}_${v}_$rep.json 2>/dev/null; done
done; done
for f in $out/*.json; do python -c "
import json; d=json.loads(open('$f').read().strip().splitlines()[-1]); r=d['roofline']; s=d['samples']; print('$f'.split('/')[-1], 'value', round(d['value']/1e9,3), 'kfrac', round(r['kernel_frac'],4), 'sust', round(d['sustained']['value']/1e9,3), 'median', round(s['median']/1e9,3), 'p10', round(s['p10']/1e9,3), 'min', round(s['min']/1e9,3), 'min/med', round(s['min_over_median'],3))"; done | tee $out/summary.txt
