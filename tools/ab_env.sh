# usage: ab_env.sh <outdir> <VAR> <value...>   (interleaved bench.py runs of one library under several values of an environment variable)
out=$1; var=$2; shift 2; mkdir -p $out
for rep in 1 2; do for v in "$@"; do
  env $var=$v timeout 200 python bench.py --no-cpu-baseline > $out/default_${var}_${v}_$rep.json 2>/dev/null
  env $var=$v timeout 200 python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $out/T20_${var}_${v}_$rep.json 2>/dev/null
done; done
for f in $out/*.json; do python -c "
import json; d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f'.split('/')[-1], round(d['value']/1e9,3), round(d['roofline']['frac'],4), round(d['sustained']['value']/1e9,3))"; done
