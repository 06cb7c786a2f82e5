#!/bin/bash
# GPU box: per-launch durations of the step engine and the refill kernels with a subset of the refill kernels skipped (timing only)
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
for skip in ${SKIPS:-0 13 14 11 7}; do
  out=gpurun_out/refill_alone_$skip; rm -rf $out; mkdir -p $out
  SKIP=$skip rocprofv3 --kernel-trace --stats -d $out -o runc -- python3 tools/refill_alone.py > $out/log.txt 2>&1
  echo "== skip mask $skip"; python tools/timeline.py $out/runc_results.db 3 | grep -E "refill|engine" | awk '{print $3, $5}' | sort -k2 | awk '{a[$2]=a[$2]" "$1} END {for (k in a) print k, a[k]}'
done
