# usage: ab_libs2.sh <outdir> <reps> <lib...>   (interleaved bench of several libraries on one box, kernel-only benches: no step path / small-n)
out=$1; reps=$2; shift; shift; mkdir -p $out
for rep in $(seq 1 $reps); do for lib in "$@"; do name=$(basename $lib .so)
  BALATRO_MI355X_LIB=$lib timeout 200 python bench.py --no-cpu-baseline --no-step-path --no-small-n > $out/default_${name}_$rep.json 2>/dev/null
  BALATRO_MI355X_LIB=$lib timeout 200 python bench.py --no-cpu-baseline --no-step-path --no-small-n --steps 20 --warmup 5 > $out/T20_${name}_$rep.json 2>/dev/null
done; done
for f in $out/*.json; do python -c "
import json; d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f'.split('/')[-1], round(d['value']/1e9,3), round(d['roofline']['frac'],4), round(d['sustained']['value']/1e9,3), round(d['roofline']['mean_launch_us'],1), round(d['roofline']['refill_mean_launch_us']))"; done
