#!/bin/bash
# Development aid: alternate two environment settings of bench.py inside one GPU run (same box, interleaved).
# usage: tools/ab_env.sh "A_ENV=.." "B_ENV=.." [rounds] [extra bench args]
A="$1"; B="$2"; R="${3:-3}"; shift 3
for i in $(seq 1 $R); do
  for tag in A B; do
    if [ $tag = A ]; then E="$A"; else E="$B"; fi
    env $E python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$tag [$E]', round(d['value']/1e9,3), 'G  launch_us', round(d['roofline']['mean_launch_us'],1), 'frac', round(d['roofline']['frac'],4))"
  done
done
