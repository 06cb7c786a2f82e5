#!/usr/bin/env python3
"""Run bench.py with the given extra args / env and print the few numbers an A/B needs on one line."""
import json, os, subprocess, sys
args = sys.argv[1:]
out = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "bench.py"), "--no-cpu-baseline", "--no-step-path"] + args,
                     capture_output=True, text=True)
line = [l for l in out.stdout.splitlines() if l.startswith("{")]
if not line:
    print("FAILED", out.stdout[-400:], out.stderr[-800:]); sys.exit(1)
j = json.loads(line[-1]); r = j["roofline"]
tag = " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("BG_"))
print(f"{tag:40s} value {j['value']/1e9:6.3f} G  kernel {r['mean_launch_us']:8.1f} us x{r['launches']:3d}  frac {r['frac']:.3f}  refill {r.get('refill_mean_launch_us', 0):7.1f} us  T {j['config']['fused_steps_per_launch']:.0f}")
