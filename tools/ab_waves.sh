#!/bin/bash
# GPU box: launch time of T fused steps against the number of worker waves (BG_ENG_WAVES); kernel microseconds per launch
cd "$(dirname "$0")/.."
python tools/bench_brief.py --steps 20 --warmup 5 > /dev/null
for T in 4 10 20 40 80 160 372; do
  line="T=$T:"
  for w in 4 5 6 7; do
    us=$(BG_ENG_WAVES=$w python tools/bench_brief.py --chunk $T --steps $((T*6)) --warmup $((T*2)) | sed 's/.*kernel *\([0-9.]*\) us.*/\1/')
    line="$line  w$w $us"
  done
  echo "$line"
done
