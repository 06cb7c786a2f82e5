#!/bin/bash
# early answers (parked lanes finish from their packed state): parity, batch cycles, interleaved A/B against HEAD and against the same binary with the flag off
out=gpurun_out/r05t; mkdir -p $out; export TMPDIR=/tmp
(BG_E3_EARLY=1 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "golden_trace or every_engine or full_size_slice or configs1 or fused_rollout_vs_oracle or card_states or consumables_rollout or immolate or many_short or curriculum or shop_stream" > $out/gpu_tests_early.txt 2>&1; echo rc=$? >> $out/gpu_tests_early.txt); tail -3 $out/gpu_tests_early.txt
for e in 0 1; do for T in 20 372; do BG_E3_EARLY=$e BALATRO_MI355X_LIB=build/variants/e3t.so N=65536 T=$T WARM=$T timeout 300 python tools/e3_timing.py 2>&1 | grep -E "launch|SERVICE p|SERVICE o|OWNER \(" | sed "s/^/early=$e T=$T: /" | tee -a $out/e3t.txt; done; done
for rep in 1 2 3 4; do
  BALATRO_MI355X_LIB=build/variants/base.so timeout 200 python bench.py --no-cpu-baseline --no-step-path --no-small-n --steps 20 --warmup 5 > $out/T20_base_$rep.json 2>/dev/null
  BG_E3_EARLY=0 timeout 200 python bench.py --no-cpu-baseline --no-step-path --no-small-n --steps 20 --warmup 5 > $out/T20_early0_$rep.json 2>/dev/null
  BG_E3_EARLY=1 timeout 200 python bench.py --no-cpu-baseline --no-step-path --no-small-n --steps 20 --warmup 5 > $out/T20_early1_$rep.json 2>/dev/null
done
for rep in 1 2; do
  BALATRO_MI355X_LIB=build/variants/base.so timeout 200 python bench.py --no-cpu-baseline --no-step-path --no-small-n > $out/default_base_$rep.json 2>/dev/null
  BG_E3_EARLY=0 timeout 200 python bench.py --no-cpu-baseline --no-step-path --no-small-n > $out/default_early0_$rep.json 2>/dev/null
  BG_E3_EARLY=1 timeout 200 python bench.py --no-cpu-baseline --no-step-path --no-small-n > $out/default_early1_$rep.json 2>/dev/null
done
for f in $out/*.json; do python -c "
import json; d=json.loads(open('$f').read().strip().splitlines()[-1]); r=d['roofline']; print('$f'.split('/')[-1], 'value', round(d['value']/1e9,3), 'kfrac', round(r['kernel_frac'],4), 'sust', round(d['sustained']['value']/1e9,3), 'launch_us', round(r['mean_launch_us'],1), 'median', round(d['samples']['median']/1e9,3))"; done
