#!/usr/bin/env python3
"""Turn the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, as MI355X_MICROARCH.md prescribes) of
`bench.py` into profiles/<tag>_hbm_traffic.json: HBM bytes per launch / per env-step of the rollout kernel.

usage: hbm_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <envs> <fused_steps> <out.json> [note]
"""
import csv, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from balatro_gym_amd import _native as nat


KERNELS = ("bg_engine3_kernel", "bg_engine_kernel")  # the step engine (packed-record rollouts: bg_engine3.h)
# bench.py only quotes a measurement taken on the device code it runs: the sha256 of the library's .hip_fatbin section (run this script
# with the same BALATRO_MI355X_LIB / product library the profiled command used)


def mean_counter(path, counter):
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter]
    for kernel in KERNELS:
        vals = [float(r["Counter_Value"]) for r in rows if kernel in r["Kernel_Name"]]
        if vals:
            full = [r["Kernel_Name"] for r in rows if kernel in r["Kernel_Name"]][0]
            vals = vals[len(vals) // 4:]  # steady state: drop the first quarter (first launches start from cold rings)
            return sum(vals) / len(vals), len(vals), kernel, full
    raise SystemExit(f"no launch of {KERNELS} in {path}")


def main():
    fcsv, wcsv, envs, fused, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
    note = sys.argv[6] if len(sys.argv) > 6 else ""
    f, nf, kernel, full = mean_counter(fcsv, "FETCH_SIZE")
    w, nw, kernel_w, _ = mean_counter(wcsv, "WRITE_SIZE")
    assert kernel == kernel_w
    hbm = 2 * f * 1024 + w * 1024
    alg = 330 + 10 + 2 * 192 / fused
    json.dump({
        "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py "
                  "--no-cpu-baseline, " + full.split("(")[0] + ", mean over steady-state launches. " + note,
        "kernel": kernel, "device_code_sha": nat.device_code_signature(), "envs": envs, "fused_steps_per_launch": fused, "launches_averaged": [nf, nw],
        "FETCH_SIZE_KB_per_launch": f, "WRITE_SIZE_KB_per_launch": w,
        "correction": "gfx950 FETCH_SIZE counts 1/2 of wide (16 B/lane) coalesced reads (MI355X_MICROARCH.md HBM): fetch "
                      "bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE taken as reported (* 1024)",
        "hbm_bytes_per_launch": hbm, "hbm_bytes_per_env_step": hbm / (envs * fused),
        "algorithmic_bytes_per_env_step": alg,
    }, open(out, "w"), indent=1)
    print(open(out).read())


if __name__ == "__main__":
    main()
