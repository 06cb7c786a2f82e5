#!/bin/bash
# GPU box: everything a round commits under profiles/rNN_final/ for the shipped library -- the bench lines (default shape, the driver's shape), rocprofv3
# --kernel-trace --stats of the DRIVER'S EXACT COMMAND and of the default command, a run made of 20-step launches only (kernel medians), the two PMC passes per
# launch shape turned into *_hbm_traffic.json (keyed on the library's build signature), the SQ counters of bg_engine3_kernel at both launch shapes, the refill in
# pieces against the whole refill at the driver's shape.   usage: tools/round_artefacts.sh <tag> [round label]   (outputs under gpurun_out/<tag>/)
# (the GPU suite and the stress runs are steps of tools/gpu_box.sh: tests, stress)
set -u
tag="$1"; label="${2:-Round-6 build.}"; out="gpurun_out/$tag"; mkdir -p "$out"
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
python bench.py > "$out/bench_default.json" 2> "$out/bench_default.err"
python bench.py --gpus 1 --steps 20 --warmup 5 > "$out/bench_driver_shape.json" 2> "$out/bench_driver_shape.err"
# (under rocprofv3 the program itself follows `--`: no env / bash -c hop)
rocprofv3 --kernel-trace --stats -d "$out/prof_driver" -o runc -- python3 bench.py --gpus 1 --steps 20 --warmup 5 > "$out/bench_driver_prof.json" 2> "$out/bench_driver_prof.err"
python tools/rocpd_summary.py "$(find $out/prof_driver -name '*.db' | head -1)" "$out/driver_cmd_kernel_stats.txt" > /dev/null 2>&1
python tools/driver_cmd_short.py "$(find $out/prof_driver -name '*.db' | head -1)" > "$out/driver_cmd_short_launches.txt" 2>&1
rocprofv3 --kernel-trace --stats -d "$out/prof_default" -o runc -- python3 bench.py --no-cpu-baseline --no-step-path > "$out/bench_default_prof.json" 2> "$out/bench_default_prof.err"
python tools/rocpd_summary.py "$(find $out/prof_default -name '*.db' | head -1)" "$out/kernel_stats.txt" > /dev/null 2>&1
rocprofv3 --kernel-trace --stats -d "$out/prof_t20" -o runc -- python3 bench.py --no-cpu-baseline --no-step-path --no-small-n --internal-warmup-launches 0 --chunk 20 --steps 2000 --warmup 400 --samples 0 > "$out/bench_t20_prof.json" 2> "$out/bench_t20_prof.err"
python tools/kernel_medians.py "$(find $out/prof_t20 -name '*.db' | head -1)" > "$out/t20_only_kernel_medians.txt" 2>&1
rm -rf "$out/prof_driver" "$out/prof_default" "$out/prof_t20"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$out/pmc_$c" -o runc -- python3 bench.py --no-cpu-baseline --no-step-path --no-small-n --steps 3720 --warmup 3720 --samples 0 > "$out/pmc_$c.json" 2> "$out/pmc_$c.err"
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$out/pmc20_$c" -o runc -- python3 bench.py --no-cpu-baseline --no-step-path --no-small-n --internal-warmup-launches 0 --chunk 20 --steps 2000 --warmup 400 --samples 0 > "$out/pmc20_$c.json" 2> "$out/pmc20_$c.err"
done
python tools/hbm_traffic.py "$out/pmc_FETCH_SIZE/runc_counter_collection.csv" "$out/pmc_WRITE_SIZE/runc_counter_collection.csv" 65536 372 "$out/hbm_traffic.json" "$label" > /dev/null 2>&1
python tools/hbm_traffic.py "$out/pmc20_FETCH_SIZE/runc_counter_collection.csv" "$out/pmc20_WRITE_SIZE/runc_counter_collection.csv" 65536 20 "$out/T20_hbm_traffic.json" "$label  The driver's launch shape (20 fused steps per launch)." > /dev/null 2>&1
for d in pmc_FETCH_SIZE pmc_WRITE_SIZE pmc20_FETCH_SIZE pmc20_WRITE_SIZE; do
  (head -1 "$out/$d/runc_counter_collection.csv"; grep "bg_engine" "$out/$d/runc_counter_collection.csv" | head -400) > "$out/$d.csv" 2>/dev/null
  rm -rf "$out/$d"
done
# SQ counters of the engine kernel alone on the GPU (synchronous refill), both launch shapes
tools/sq_counters.sh "$tag/sq372" 372 > /dev/null 2>&1; python tools/sq_summary.py "$tag/sq372" bg_engine3_kernel > "$out/sq_counters.txt" 2>&1
tools/sq_counters.sh "$tag/sq20" 20 > /dev/null 2>&1; python tools/sq_summary.py "$tag/sq20" bg_engine3_kernel > "$out/sq_counters_T20.txt" 2>&1
rm -rf "$out/sq372" "$out/sq20"
# the refill in pieces (the default) against the whole refill beside the next launches (BG_REFILL_SLICED=0), interleaved, at the driver's shape
for rep in 1 2 3; do for sl in 1 0; do
  BG_REFILL_SLICED=$sl python bench.py --no-cpu-baseline --no-step-path --no-small-n --steps 20 --warmup 5 --samples 60 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; s=d['samples']
print('BG_REFILL_SLICED=$sl rep $rep: value %.3f G  wall frac %.4f  kernel frac %.4f  sustained %.3f G  samples median %.3f  p10 %.3f  min %.3f  min/median %.3f' % (d['value']/1e9, r['frac'], r['kernel_frac'], d['sustained']['value']/1e9, s['median']/1e9, s['p10']/1e9, s['min']/1e9, s['min_over_median']))"
done; done > "$out/refill_pieces_ab.txt" 2>&1
python - "$out" <<'PY'
import json, os, sys
out = sys.argv[1]
for name in ("bench_default", "bench_driver_shape"):
    try:
        d = json.loads(open(os.path.join(out, name + ".json")).read().strip().splitlines()[-1]); r = d["roofline"]
        print(name, "value", round(d["value"] / 1e9, 3), "frac", round(r["frac"], 4), "kernel_frac", round(r["kernel_frac"], 4), "sustained", round(d["sustained"]["value"] / 1e9, 3),
              "samples min/median", round(d["samples"]["min_over_median"], 3), "small_n", d.get("small_n", {}).get("value"), "traffic", r["traffic"])
    except Exception as ex:
        print(name, "failed:", ex)
for name in ("hbm_traffic", "T20_hbm_traffic"):
    try:
        d = json.load(open(os.path.join(out, name + ".json"))); print(name, d["hbm_bytes_per_env_step"], d["device_code_sha"])
    except Exception as ex:
        print(name, "failed:", ex)
PY
head -12 "$out/driver_cmd_kernel_stats.txt"; head -30 "$out/sq_counters.txt"
