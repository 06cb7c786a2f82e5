#!/bin/bash
# timeline of single 20-step launches (synchronised one by one, as bench.py's samples) around a look-ahead refill
out=gpurun_out/r05v; mkdir -p $out; export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$out/prof -o run -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-step-path --no-small-n --steps 20 --warmup 5 --samples 60 > $GRAFT_REPO_ROOT/$out/bench.json 2> $GRAFT_REPO_ROOT/$out/bench.err
cd $GRAFT_REPO_ROOT
f=$(find $out/prof -name "*kernel_trace.csv" | head -1)
python tools/timeline_refill.py $f > $out/timeline.txt 2>&1
head -60 $out/timeline.txt
