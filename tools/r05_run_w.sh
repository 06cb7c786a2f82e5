#!/bin/bash
# the refill in pieces: parity, then interleaved A/B at the driver's shape, then a kernel-trace timeline
out=gpurun_out/r05w; mkdir -p $out; export TMPDIR=/tmp
(timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "sliced_refill or many_short or shallow_rings or golden_trace or fused_rollout_vs_oracle" > $out/gpu_tests.txt 2>&1; echo rc=$? >> $out/gpu_tests.txt); tail -3 $out/gpu_tests.txt
for rep in 1 2 3; do
  for sl in 0 1; do
    BG_REFILL_SLICED=$sl timeout 200 python bench.py --no-cpu-baseline --no-step-path --no-small-n --steps 20 --warmup 5 --samples 60 > $out/T20_sliced${sl}_$rep.json 2>/dev/null
  done
  BG_REFILL_SLICED=1 BG_REFILL_PARTS=6,3,3,12 timeout 200 python bench.py --no-cpu-baseline --no-step-path --no-small-n --steps 20 --warmup 5 --samples 60 > $out/T20_sliced1fine_$rep.json 2>/dev/null
done
for sl in 0 1; do BG_REFILL_SLICED=$sl timeout 200 python bench.py --no-cpu-baseline --no-step-path --no-small-n > $out/default_sliced${sl}.json 2>/dev/null; done
for f in $out/*.json; do python -c "
import json; d=json.loads(open('$f').read().strip().splitlines()[-1]); r=d['roofline']; s=d['samples']; print('$f'.split('/')[-1], 'value', round(d['value']/1e9,3), 'kfrac', round(r['kernel_frac'],4), 'sust', round(d['sustained']['value']/1e9,3), 'median', round(s['median']/1e9,3), 'p10', round(s['p10']/1e9,3), 'min', round(s['min']/1e9,3), 'min/med', round(s['min_over_median'],3))"; done | tee $out/summary.txt
cd /tmp
BG_REFILL_SLICED=1 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$out/prof -o run -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-step-path --no-small-n --steps 20 --warmup 5 --samples 60 > $GRAFT_REPO_ROOT/$out/bench_prof.json 2> $GRAFT_REPO_ROOT/$out/bench_prof.err
