#!/bin/bash
# development helper (GPU box): alternate the product library and a variant in ONE run (boxes differ by several per cent)
v=$1; n=${2:-3}
for i in $(seq $n); do
  python bench.py --no-cpu-baseline | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('head   ', round(d['value']/1e6,1), round(d['roofline']['mean_launch_us'],1))"
  BALATRO_MI355X_LIB=$PWD/balatro_gym_amd/variants/$v.so python bench.py --no-cpu-baseline | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(d['value']/1e6,1), round(d['roofline']['mean_launch_us'],1))"
done
