#!/bin/bash
# pieces up to half a refill period, this launch's share by the launches left in the period: parity, then every launch length against the shipped library of the morning
out=gpurun_out/r05ah; mkdir -p $out; export TMPDIR=/tmp
(timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "sliced_refill or many_short or shallow_rings" > $out/gpu_tests.txt 2>&1; echo rc=$? >> $out/gpu_tests.txt); tail -3 $out/gpu_tests.txt
for rep in 1 2; do for T in 20 48 100 180 372; do for sl in 1 0; do
  extra="--steps $T --warmup 5 --samples 40"; [ $T = 372 ] && extra=""
  BG_REFILL_SLICED=$sl timeout 300 python bench.py --no-cpu-baseline --no-step-path --no-small-n $extra > $out/T${T}_sliced${sl}_$rep.json 2>/dev/null
done; done; done
for f in $out/*.json; do python -c "
import json; d=json.loads(open('$f').read().strip().splitlines()[-1]); r=d['roofline']; s=d['samples']; print('$f'.split('/')[-1], 'value', round(d['value']/1e9,3), 'kfrac', round(r['kernel_frac'],4), 'launch_us', round(r['mean_launch_us'],1), 'sust', round(d['sustained']['value']/1e9,3), 'median', round(s['median']/1e9,3), 'p10', round(s['p10']/1e9,3), 'min', round(s['min']/1e9,3), 'min/med', round(s['min_over_median'],3))"; done | tee $out/summary.txt
