#!/usr/bin/env python3
"""The SHORT launches of `python3 bench.py --gpus 1 --steps 20 --warmup 5` (the driver's command) in a rocprofv3 rocpd database, in issue order:
the 5-step warm-up launch, THE timed 20-step launch, the 30 `samples` launches (each between two synchronisations) and the 20 back-to-back `sustained`
ones -- so that bench.py's HIP-event figures (roofline.mean_launch_us = the timed launch, samples.mean_launch_us) have the profiler's number for the SAME
launches beside them (the --stats average over all 180 launches of the kernel mixes them with the 128 full-depth warm-up launches).
usage: driver_cmd_short.py <rocpd .db>"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
eng = [(s, (e - s) / 1e3) for n, s, e in db.execute("select name,start,end from kernels order by start") if "bg_engine3_kernel<false, false, 4, 1, 3>" in n]
short = [d for _, d in eng if d < 1000.0]
print(f"bg_engine3_kernel<false,false,4,1,3>: {len(eng)} launches, {len(eng) - len(short)} of the internal full-depth warm-up (mean {sum(d for _, d in eng if d >= 1000.0) / max(1, len(eng) - len(short)):.1f} us), {len(short)} short ones")
if len(short) >= 52:
    warm, timed, samples, sust = short[0], short[1], short[2:32], short[32:52]
    med = lambda v: sorted(v)[len(v) // 2]
    print(f"  5-step warm-up launch                  {warm:8.1f} us")
    print(f"  THE TIMED 20-step launch               {timed:8.1f} us   (bench.py roofline.mean_launch_us of the same command)")
    print(f"  30 samples (a synchronisation between) median {med(samples):8.1f} us  mean {sum(samples) / len(samples):8.1f} us  (bench.py samples.mean_launch_us)")
    print(f"  20 sustained (back to back)            median {med(sust):8.1f} us  mean {sum(sust) / len(sust):8.1f} us  (a refill piece beside each AND the tail of the one before)")
else:
    print("  durations (us):", " ".join(f"{d:.1f}" for d in short))
