#!/bin/bash
# service-wave batch thresholds at the driver's shape and at 372 steps (BG_E3_TH requests or BG_E3_WAIT x 10 ns, whichever first)
out=gpurun_out/r05e; mkdir -p $out; export TMPDIR=/tmp
for rep in 1 2; do for cfg in "2147483647 0" "32 300" "32 600" "48 600" "48 1200" "64 1000" "24 200"; do set -- $cfg
  BG_E3_TH=$1 BG_E3_WAIT=$2 timeout 200 python bench.py --no-cpu-baseline --no-step-path --no-small-n --steps 20 --warmup 5 > $out/T20_th$1_w$2_$rep.json 2>/dev/null
done; done
for cfg in "2147483647 0" "32 300" "48 600"; do set -- $cfg
  BG_E3_TH=$1 BG_E3_WAIT=$2 timeout 200 python bench.py --no-cpu-baseline --no-step-path --no-small-n > $out/default_th$1_w$2.json 2>/dev/null
done
for f in $out/*.json; do python -c "
import json; d=json.loads(open('$f').read().strip().splitlines()[-1]); r=d['roofline']; print('$f'.split('/')[-1], 'value', round(d['value']/1e9,3), 'kfrac', round(r['kernel_frac'],4), 'sust', round(d['sustained']['value']/1e9,3), 'launch_us', round(r['mean_launch_us'],1), 'median', round(d['samples']['median']/1e9,3), 'min', round(d['samples']['min']/1e9,3))"; done
