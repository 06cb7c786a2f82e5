#!/bin/bash
out=gpurun_out/r05p; mkdir -p $out; export TMPDIR=/tmp
for rep in 1 2; do for cfg in 413 212; do
  BG_E3_CFG=$cfg timeout 200 python bench.py --no-cpu-baseline --no-step-path --no-small-n --steps 20 --warmup 5 > $out/T20_cfg${cfg}_$rep.json 2>/dev/null
  BG_E3_CFG=$cfg timeout 200 python bench.py --no-cpu-baseline --no-step-path --no-small-n > $out/default_cfg${cfg}_$rep.json 2>/dev/null
done; done
for f in $out/*.json; do python -c "
import json; d=json.loads(open('$f').read().strip().splitlines()[-1]); r=d['roofline']; print('$f'.split('/')[-1], 'value', round(d['value']/1e9,3), 'kfrac', round(r['kernel_frac'],4), 'sust', round(d['sustained']['value']/1e9,3), 'launch_us', round(r['mean_launch_us'],1), 'median', round(d['samples']['median']/1e9,3))"; done
