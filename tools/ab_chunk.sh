for i in 1 2 3; do
python bench.py --no-cpu-baseline --steps 2976 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('T128', round(d['value']/1e9,3), 'G  launch_us', round(d['roofline']['mean_launch_us'],1), 'frac', round(d['roofline']['frac'],4))"
BG_KG=97 BG_KS=193 BG_KD=192 python bench.py --no-cpu-baseline --steps 2976 --chunk 256 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('T256', round(d['value']/1e9,3), 'G  launch_us', round(d['roofline']['mean_launch_us'],1), 'frac', round(d['roofline']['frac'],4))"
BG_KG=137 BG_KS=249 BG_KD=248 python bench.py --no-cpu-baseline --steps 2976 --chunk 372 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('T372', round(d['value']/1e9,3), 'G  launch_us', round(d['roofline']['mean_launch_us'],1), 'frac', round(d['roofline']['frac'],4))"
done
