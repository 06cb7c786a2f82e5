#!/bin/bash
# GPU box: the refill in pieces with 8 (default) / 10 / 12 shop parts per period at the driver's launch shape, interleaved (BG_REFILL_PARTS is read once per handle)
out=gpurun_out/r06u_parts; mkdir -p $out
for rep in 1 2; do for parts in "2,1,2,8" "2,1,2,10" "2,1,2,12"; do
  BG_REFILL_PARTS=$parts timeout 300 python bench.py --no-cpu-baseline --no-step-path --no-small-n --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; s=d['samples']
print('parts $parts rep $rep: value %.3f G wall frac %.4f kernel frac %.4f launch %.1f us sustained %.3f G samples median %.3f p10 %.3f min/med %.3f' % (d['value']/1e9, r['frac'], r['kernel_frac'], r['mean_launch_us'], d['sustained']['value']/1e9, s['median']/1e9, s['p10']/1e9, s['min_over_median']))"
done; done > $out/parts_ab.txt 2>&1
cat $out/parts_ab.txt
