#!/bin/bash
# round 5, GPU box: suite + refill policy A/B at the driver's shape + kernel statistics of the driver's exact command and of a run made of 20-step launches only
out=gpurun_out/r05b; mkdir -p $out; export TMPDIR=/tmp
(timeout 1200 python -m pytest tests -m gpu -x -q > $out/gpu_tests.txt 2>&1; echo rc=$? >> $out/gpu_tests.txt); tail -3 $out/gpu_tests.txt
for rep in 1 2; do for m in 0 16; do
  BG_REFILL_MIN=$m timeout 200 python bench.py --no-cpu-baseline --no-step-path --no-small-n --steps 20 --warmup 5 > $out/T20_min${m}_$rep.json 2>/dev/null
done; done
BG_REFILL_MIN=16 timeout 200 python bench.py --no-cpu-baseline --no-step-path --no-small-n > $out/default_min16.json 2>/dev/null
for f in $out/*.json; do python -c "
import json; d=json.loads(open('$f').read().strip().splitlines()[-1]); r=d['roofline']; print('$f'.split('/')[-1], 'value', round(d['value']/1e9,3), 'frac', round(r['frac'],4), 'kfrac', round(r['kernel_frac'],4), 'sust', round(d['sustained']['value']/1e9,3), 'launch_us', round(r['mean_launch_us'],1), 'median', round(d['samples']['median']/1e9,3), 'min', round(d['samples']['min']/1e9,3), 'refill_us', round(r['refill_kernel_us_in_timed_region'],1))"; done
rocprofv3 --kernel-trace --stats -d $out/prof_driver -o runc -- python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver_prof.json 2> $out/bench_driver_prof.err
python tools/rocpd_summary.py $(find $out/prof_driver -name "*.db" | head -1) $out/driver_cmd_kernel_stats.txt > /dev/null 2>&1
rocprofv3 --kernel-trace --stats -d $out/prof_t20 -o runc -- python3 bench.py --no-cpu-baseline --no-step-path --no-small-n --internal-warmup-launches 0 --chunk 20 --steps 2000 --warmup 400 --samples 0 > $out/bench_t20_prof.json 2> $out/bench_t20_prof.err
python tools/rocpd_summary.py $(find $out/prof_t20 -name "*.db" | head -1) $out/t20_only_kernel_stats.txt > /dev/null 2>&1
python tools/kernel_medians.py $(find $out/prof_t20 -name "*.db" | head -1) > $out/t20_only_kernel_medians.txt 2>&1
rm -rf $out/prof_driver $out/prof_t20
head -14 $out/driver_cmd_kernel_stats.txt; cat $out/t20_only_kernel_medians.txt
