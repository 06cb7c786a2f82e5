#!/bin/bash
# the whole GPU suite with the refill in pieces + the moved deck-length fences; a quick look at both launch shapes
out=gpurun_out/r05ab; mkdir -p $out; export TMPDIR=/tmp
(timeout 2400 python -m pytest tests -m gpu -x -q > $out/gpu_tests.txt 2>&1; echo rc=$? >> $out/gpu_tests.txt); tail -4 $out/gpu_tests.txt
for rep in 1 2; do
  timeout 200 python bench.py --no-cpu-baseline --no-step-path --no-small-n --steps 20 --warmup 5 > $out/T20_$rep.json 2>/dev/null
  timeout 200 python bench.py --no-cpu-baseline --no-step-path --no-small-n > $out/default_$rep.json 2>/dev/null
done
for f in $out/*.json; do python -c "
import json; d=json.loads(open('$f').read().strip().splitlines()[-1]); r=d['roofline']; s=d['samples']; print('$f'.split('/')[-1], 'value', round(d['value']/1e9,3), 'frac', round(r['frac'],4), 'kfrac', round(r['kernel_frac'],4), 'sust', round(d['sustained']['value']/1e9,3), 'median', round(s['median']/1e9,3), 'min/med', round(s['min_over_median'],3))"; done | tee $out/summary.txt
