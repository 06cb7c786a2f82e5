#!/bin/bash
# GPU box: the two separate PMC passes (FETCH_SIZE / WRITE_SIZE) of bench.py for the HBM traffic of the step engine, at 372 and at 20 fused steps.
# usage: tools/pmc_traffic.sh <tag>      (outputs under gpurun_out/<tag>/; then tools/hbm_traffic.py turns the CSVs into profiles/*_hbm_traffic.json)
set -u
tag="$1"; out="gpurun_out/$tag"; mkdir -p "$out"
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$out/pmc_$c" -o runc -- python3 bench.py --no-cpu-baseline --no-step-path --samples 0 --steps 3720 --warmup 3720 > "$out/pmc_$c.json" 2> "$out/pmc_$c.err"
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$out/pmc20_$c" -o runc -- python3 bench.py --no-cpu-baseline --no-step-path --samples 0 --internal-warmup-s 0 --chunk 20 --steps 2000 --warmup 400 > "$out/pmc20_$c.json" 2> "$out/pmc20_$c.err"
done
find "$out" -name "*counter_collection.csv"
