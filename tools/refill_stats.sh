#!/bin/bash
# GPU box: durations of the refill kernels when nothing else runs beside them (synchronous refill), rocprofv3 kernel trace
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp BG_ASYNC_REFILL=0
out=gpurun_out/${1:-refill_stats}; mkdir -p $out
rocprofv3 --kernel-trace --stats -d $out/prof -o runc -- python3 bench.py --no-cpu-baseline --no-step-path --steps 3720 --warmup 744 > $out/bench.json 2> $out/bench.err
python tools/rocpd_summary.py $out/prof/runc_results.db | head -12
