for rb in 512 2048 8192 32768; do
BG_REFILL_BLOCKS=$rb BG_ROLLOUT_V=3 BG_TH_PLAY=1 BG_TH_OTHER=1 BG_TH_READY=200 python bench.py --no-cpu-baseline --warmup 7440 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('v3 refill_blocks $rb', round(d['value']/1e9,3), 'G rollout_us', round(d['roofline']['mean_launch_us'],1), 'refill_us', round(d['roofline']['refill_mean_launch_us'],1))"
done
BG_REFILL_BLOCKS=8192 BG_ROLLOUT_V=2 python bench.py --no-cpu-baseline --warmup 7440 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('v2 refill_blocks 8192', round(d['value']/1e9,3), 'G rollout_us', round(d['roofline']['mean_launch_us'],1), 'refill_us', round(d['roofline']['refill_mean_launch_us'],1))"
BG_ROLLOUT_V=2 python bench.py --no-cpu-baseline --warmup 7440 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('v2 default', round(d['value']/1e9,3), 'G rollout_us', round(d['roofline']['mean_launch_us'],1), 'refill_us', round(d['roofline']['refill_mean_launch_us'],1))"
