#!/bin/bash
# scheduling experiments at the driver's shape (and a check at 372 steps): queue preference of a free service wave, owner waves that sleep after a thin iteration
out=gpurun_out/r05f; mkdir -p $out; export TMPDIR=/tmp
for rep in 1 2; do for cfg in "0 0" "1 0" "2 0" "3 0" "0 1" "0 3" "0 8"; do set -- $cfg
  BG_E3_PREF=$1 BG_E3_OSLEEP=$2 timeout 200 python bench.py --no-cpu-baseline --no-step-path --no-small-n --steps 20 --warmup 5 > $out/T20_pref$1_osleep$2_$rep.json 2>/dev/null
done; done
for cfg in "0 0" "1 0" "3 0" "0 3"; do set -- $cfg
  BG_E3_PREF=$1 BG_E3_OSLEEP=$2 timeout 200 python bench.py --no-cpu-baseline --no-step-path --no-small-n > $out/default_pref$1_osleep$2.json 2>/dev/null
done
for f in $out/*.json; do python -c "
import json; d=json.loads(open('$f').read().strip().splitlines()[-1]); r=d['roofline']; print('$f'.split('/')[-1], 'value', round(d['value']/1e9,3), 'kfrac', round(r['kernel_frac'],4), 'sust', round(d['sustained']['value']/1e9,3), 'launch_us', round(r['mean_launch_us'],1), 'median', round(d['samples']['median']/1e9,3), 'min', round(d['samples']['min']/1e9,3))"; done
