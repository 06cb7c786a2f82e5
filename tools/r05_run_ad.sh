#!/bin/bash
# the two committed bench lines (roofline.traffic from profiles/r05*_hbm_traffic.json of this signature; step_path over 800 steps) + the step-path test on the shipped library
out=gpurun_out/r05ak; mkdir -p $out; export TMPDIR=/tmp
(timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "step_path_across or sliced_refill or deck_length" > $out/gpu_tests_new.txt 2>&1; echo rc=$? >> $out/gpu_tests_new.txt); tail -3 $out/gpu_tests_new.txt
python bench.py > $out/bench_default.json 2> $out/bench_default.err
python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver_shape.json 2> $out/bench_driver_shape.err
for f in $out/*.json; do python -c "
import json; d=json.loads(open('$f').read().strip().splitlines()[-1]); r=d['roofline']; s=d['samples']; print('$f'.split('/')[-1], 'value', round(d['value']/1e9,3), 'frac', round(r['frac'],4), 'kfrac', round(r['kernel_frac'],4), 'traffic', r['traffic'], 'sust', round(d['sustained']['value']/1e9,3), 'median', round(s['median']/1e9,3), 'min/med', round(s['min_over_median'],3), 'small', round(d['small_n']['value']/1e9,3), {k:round(v['value']/1e9,3) for k,v in d['step_path'].items() if isinstance(v,dict)})"; done
