#!/bin/bash
out=gpurun_out/r05ai; mkdir -p $out; export TMPDIR=/tmp
(timeout 2400 python -m pytest tests -m gpu -x -q > $out/gpu_tests.txt 2>&1; echo rc=$? >> $out/gpu_tests.txt); tail -3 $out/gpu_tests.txt
for rep in 1 2; do for T in 100 180; do
  timeout 300 python bench.py --no-cpu-baseline --no-step-path --no-small-n --steps $T --warmup 5 --samples 40 > $out/T${T}_$rep.json 2>/dev/null
done; done
for f in $out/*.json; do python -c "
import json; d=json.loads(open('$f').read().strip().splitlines()[-1]); r=d['roofline']; s=d['samples']; print('$f'.split('/')[-1], 'value', round(d['value']/1e9,3), 'kfrac', round(r['kernel_frac'],4), 'launch_us', round(r['mean_launch_us'],1), 'sust', round(d['sustained']['value']/1e9,3), 'median', round(s['median']/1e9,3), 'p10', round(s['p10']/1e9,3), 'min', round(s['min']/1e9,3), 'min/med', round(s['min_over_median'],3))"; done | tee $out/summary.txt
