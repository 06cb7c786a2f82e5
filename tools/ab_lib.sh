#!/bin/bash
# Development aid: alternate the product library and a variant build (tools/build_variant.sh) inside one GPU run.
# usage: tools/ab_lib.sh <variant.so> [rounds] [extra bench args]
V="$1"; R="${2:-3}"; shift 2
for i in $(seq 1 $R); do
  for tag in product variant; do
    if [ $tag = variant ]; then export BALATRO_MI355X_LIB="$V"; else unset BALATRO_MI355X_LIB; fi
    python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$tag', round(d['value']/1e9,3), 'G  launch_us', round(d['roofline']['mean_launch_us'],1), 'frac', round(d['roofline']['frac'],4))"
  done
done
