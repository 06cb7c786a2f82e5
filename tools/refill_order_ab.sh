#!/bin/bash
# GPU box: per-kernel durations of the refill beside the engine for BG_REFILL_ORDER = 0 (shop seeding beside the deck / seed-ring / block
# kernels) and 1 (after them), rocprofv3 kernel trace; then the same two settings interleaved without the profiler.
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
out=gpurun_out/${1:-refill_order}; mkdir -p $out
for o in 0 1 2; do
  export BG_REFILL_ORDER=$o
  timeout 600 rocprofv3 --kernel-trace --stats -d $out/prof$o -o runc -- python3 bench.py --no-cpu-baseline --no-step-path --samples 2 --steps 7440 --warmup 744 > $out/bench$o.json 2> $out/bench$o.err
  echo "== BG_REFILL_ORDER=$o"; python tools/rocpd_summary.py $out/prof$o/runc_results.db | head -10
done
for i in 1 2; do for o in 0 1 2; do export BG_REFILL_ORDER=$o; timeout 300 python tools/bench_brief.py --samples 2; done; done
for o in 0 1 2; do export BG_REFILL_ORDER=$o; timeout 300 python tools/bench_brief.py --samples 2 --steps 20 --warmup 5; done
