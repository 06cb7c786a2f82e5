#!/bin/bash
# GPU box: produce the artefacts a round commits under profiles/ -- default bench line, rocprofv3 kernel stats of the same
# command, and the two separate PMC passes (FETCH_SIZE / WRITE_SIZE) for the HBM traffic of the rollout kernel.
# usage: tools/round_artefacts.sh <tag>      (outputs under gpurun_out/<tag>/)
set -u
tag="$1"; out="gpurun_out/$tag"; mkdir -p "$out"
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
python bench.py > "$out/bench_default.json" 2> "$out/bench_default.err"
python bench.py --no-cpu-baseline --no-step-path --envs-per-gpu 4096 > "$out/bench_4096.json" 2> "$out/bench_4096.err"
rocprofv3 --kernel-trace --stats -d "$out/prof" -o runc -- python3 bench.py --no-cpu-baseline --no-step-path > "$out/bench_prof.json" 2> "$out/bench_prof.err"
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-step-path > "$out/bench_driver_shape.json" 2> "$out/bench_driver_shape.err"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$out/pmc_$c" -o runc -- python3 bench.py --no-cpu-baseline --no-step-path --steps 3720 --warmup 3720 > "$out/pmc_$c.json" 2> "$out/pmc_$c.err"
  # the driver's launch shape (--steps 20): every launch of this pass fuses 20 steps
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$out/pmc20_$c" -o runc -- python3 bench.py --no-cpu-baseline --no-step-path --internal-warmup-s 0 --chunk 20 --steps 2000 --warmup 400 > "$out/pmc20_$c.json" 2> "$out/pmc20_$c.err"
done
find "$out" -name "*.db" -o -name "*counter_collection.csv" | head
