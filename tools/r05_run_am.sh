#!/bin/bash
# the copy-out's one-group path for sparse owner iterations (<= 8 records): parity with it, then interleaved A/B against the shipped library: driver's shape, default shape (+ small_n)
out=gpurun_out/r05ar; mkdir -p $out; export TMPDIR=/tmp
(BALATRO_MI355X_LIB=build/variants/hoist.so timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "golden_trace or every_engine or full_size_slice or configs1 or fused_rollout_vs_oracle or card_states or many_short or sliced_refill or packed_record" > $out/gpu_tests_sparse.txt 2>&1; echo rc=$? >> $out/gpu_tests_sparse.txt); tail -3 $out/gpu_tests_sparse.txt
for rep in 1 2 3; do for v in shipped hoist; do
  lib=build/variants/$v.so; [ $v = shipped ] && lib=balatro_gym_amd/libbalatro_mi355x.so
  BALATRO_MI355X_LIB=$lib timeout 200 python bench.py --no-cpu-baseline --no-step-path --no-small-n --steps 20 --warmup 5 > $out/T20_${v}_$rep.json 2>/dev/null
  BALATRO_MI355X_LIB=$lib timeout 300 python bench.py --no-cpu-baseline --no-step-path > $out/default_${v}_$rep.json 2>/dev/null
done; done
for f in $out/*.json; do python -c "
import json; d=json.loads(open('$f').read().strip().splitlines()[-1]); r=d['roofline']; s=d['samples']; sn=d.get('small_n'); print('$f'.split('/')[-1], 'value', round(d['value']/1e9,3), 'kfrac', round(r['kernel_frac'],4), 'launch_us', round(r['mean_launch_us'],1), 'sust', round(d['sustained']['value']/1e9,3), 'median', round(s['median']/1e9,3), 'small_n', round(sn['value']/1e9,3) if sn else None)"; done | tee $out/summary.txt
