#!/bin/bash
# bg_step / bg_step_many with the refill overlapped (in pieces beside the step launches) against the synchronous refill on the stream (BG_ASYNC_REFILL=0): parity, then 800 steps
out=gpurun_out/r05aj; mkdir -p $out; export TMPDIR=/tmp
(timeout 2400 python -m pytest tests -m gpu -x -q > $out/gpu_tests.txt 2>&1; echo rc=$? >> $out/gpu_tests.txt); tail -3 $out/gpu_tests.txt
for rep in 1 2; do for as in 1 0; do
  BG_ASYNC_REFILL=$as timeout 600 python bench.py --no-cpu-baseline --no-small-n --samples 0 > $out/steppath_async${as}_$rep.json 2>/dev/null
done; done
for f in $out/*.json; do python -c "
import json; d=json.loads(open('$f').read().strip().splitlines()[-1]); sp=d['step_path']; print('$f'.split('/')[-1], 'value', round(d['value']/1e9,3), {k:(round(v['value']/1e9,3), round(v['ms_per_step']*1e3,2), round(v['kernel_ms_per_step']*1e3,2)) for k,v in sp.items() if isinstance(v,dict)})"; done | tee $out/summary.txt
