#!/bin/bash
# GPU box: A/B of the two lane mappings (tools/ab_lanes.py) + rocprofv3 kernel stats of the same run
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
out=gpurun_out/lanes; rm -rf $out; mkdir -p $out
python tools/ab_lanes.py --out $out/ab_lanes.json
rocprofv3 --kernel-trace --stats -d $out/prof -o runc -- python3 tools/ab_lanes.py > /dev/null 2> $out/prof.err
python tools/rocpd_summary.py $out/prof/runc_results.db | grep -E "classify|score_hand|score_seed|#" > $out/kernel_stats.txt; cat $out/kernel_stats.txt
