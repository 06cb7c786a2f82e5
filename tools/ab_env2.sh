# usage: ab_env2.sh <outdir> "<VAR=val VAR2=val2>" ...   (interleaved default-shape bench.py runs under several environments)
out=$1; shift; mkdir -p $out
for rep in 1 2; do i=0; for e in "$@"; do i=$((i+1))
  env $e timeout 200 python bench.py --no-cpu-baseline > $out/default_${i}_$rep.json 2>/dev/null
  python -c "
import json; d=json.loads(open('$out/default_${i}_$rep.json').read().strip().splitlines()[-1]); print('$e', round(d['value']/1e9,3), round(d['roofline']['frac'],4), round(d['sustained']['value']/1e9,3))"
done; done
