#!/bin/bash
# Development aid: does a longer warm-up change the measured rate (clock ramp)?
for w in 372 37200 372 37200 372 111600; do
python bench.py --no-cpu-baseline --warmup $w 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('warmup $w', round(d['value']/1e9,3), 'G  launch_us', round(d['roofline']['mean_launch_us'],1), 'frac', round(d['roofline']['frac'],4))"
done
