#!/bin/bash
# refill grid sizes at the driver's shape: do shorter-lived refill waves let the next launch's workgroups in sooner?
out=gpurun_out/r05o; mkdir -p $out; export TMPDIR=/tmp
for rep in 1 2; do for b in 4096 1024 16384 65536; do
  BG_REFILL_BLOCKS=$b timeout 200 python bench.py --no-cpu-baseline --no-step-path --no-small-n --steps 20 --warmup 5 > $out/T20_blocks${b}_$rep.json 2>/dev/null
done; done
for f in $out/*.json; do python -c "
import json; d=json.loads(open('$f').read().strip().splitlines()[-1]); r=d['roofline']; print('$f'.split('/')[-1], 'value', round(d['value']/1e9,3), 'kfrac', round(r['kernel_frac'],4), 'sust', round(d['sustained']['value']/1e9,3), 'launch_us', round(r['mean_launch_us'],1), 'median', round(d['samples']['median']/1e9,3), 'min', round(d['samples']['min']/1e9,3), 'p10', round(d['samples']['p10']/1e9,3))"; done
