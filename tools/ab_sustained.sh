#!/bin/bash
# Development aid: the driver's shape (--steps 20 --warmup 5) under alternating environment settings: value, median of 30 samples, sustained.
# usage: tools/ab_sustained.sh "A_ENV=.." "B_ENV=.." [rounds]
cd "$(dirname "$0")/.." || exit 1
A="$1"; B="$2"; R="${3:-3}"
for i in $(seq 1 $R); do
  for E in "$A" "$B"; do
    env $E timeout 300 python bench.py --no-cpu-baseline --no-step-path --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); s=d['samples']; u=d['sustained']
print('[$E]', 'value %.3f G' % (d['value']/1e9), 'median %.3f' % (s['median']/1e9), 'min %.3f' % (s['min']/1e9), 'sustained %.3f (refills inside %d)' % (u['value']/1e9, u['refill_launches_inside']))"
  done
done
