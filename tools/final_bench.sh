#!/bin/bash
# GPU box: the three bench lines a round commits (default, the driver's shape, 4 096 envs)
cd "$(dirname "$0")/.." || exit 1
out=gpurun_out/${1:-final}; mkdir -p $out
python bench.py > $out/bench_default.json 2> $out/bench_default.err
python bench.py --steps 20 --warmup 5 > $out/bench_driver_shape.json 2> $out/bench_driver_shape.err
python bench.py --no-cpu-baseline --no-step-path --envs-per-gpu 4096 > $out/bench_4096.json 2> $out/bench_4096.err
for f in default driver_shape 4096; do tail -1 $out/bench_$f.json | cut -c1-220; done
