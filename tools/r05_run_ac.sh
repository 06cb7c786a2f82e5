#!/bin/bash
# compiler scheduling strategies for the engine (same sources): -mllvm -amdgpu-sched-strategy=max-ilp / max-memory-clause, -amdgpu-schedule-metric-bias=0, against the shipped build
out=gpurun_out/r05ac; mkdir -p $out; export TMPDIR=/tmp
for rep in 1 2 3; do for v in shipped ilp memclause bias0; do
  lib=build/variants/$v.so; [ $v = shipped ] && lib=balatro_gym_amd/libbalatro_mi355x.so
  BALATRO_MI355X_LIB=$lib timeout 200 python bench.py --no-cpu-baseline --no-step-path --no-small-n --steps 20 --warmup 5 > $out/T20_${v}_$rep.json 2>/dev/null
  BALATRO_MI355X_LIB=$lib timeout 200 python bench.py --no-cpu-baseline --no-step-path --no-small-n > $out/default_${v}_$rep.json 2>/dev/null
done; done
for f in $out/*.json; do python -c "
import json; d=json.loads(open('$f').read().strip().splitlines()[-1]); r=d['roofline']; s=d['samples']; print('$f'.split('/')[-1], 'value', round(d['value']/1e9,3), 'kfrac', round(r['kernel_frac'],4), 'sust', round(d['sustained']['value']/1e9,3), 'median', round(s['median']/1e9,3), 'launch_us', round(r['mean_launch_us'],1))"; done | tee $out/summary.txt
