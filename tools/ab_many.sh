#!/bin/bash
# Development aid: the product library and several variant builds, interleaved inside one GPU run.
# usage: tools/ab_many.sh "<variant names>" <rounds> [bench args]     (variants under build/variants/<name>.so)
cd "$(dirname "$0")/.."
names="$1"; R="${2:-2}"; shift 2
for i in $(seq 1 $R); do
  for tag in product $names; do
    if [ $tag = product ]; then unset BALATRO_MI355X_LIB; else export BALATRO_MI355X_LIB="build/variants/$tag.so"; fi
    printf "%-10s" $tag; python tools/bench_brief.py "$@" | cut -c40-
  done
done
