# usage: ab.sh <outdir> <lib...>   (interleaved bench of several libraries on one box)
out=$1; shift; mkdir -p $out
for rep in 1 2; do for lib in "$@"; do name=$(basename $lib .so)
  BALATRO_MI355X_LIB=$lib timeout 200 python bench.py --no-cpu-baseline > $out/default_${name}_$rep.json 2>/dev/null
  BALATRO_MI355X_LIB=$lib timeout 200 python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $out/T20_${name}_$rep.json 2>/dev/null
done; done
for f in $out/*.json; do python -c "
import json; d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f'.split('/')[-1], round(d['value']/1e9,3), round(d['roofline']['frac'],4), round(d['sustained']['value']/1e9,3))"; done
