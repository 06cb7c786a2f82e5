#!/bin/bash
# GPU box: everything round 5 commits under profiles/r05_final/ for the shipped library -- the GPU suite, the bench lines (default shape, the driver's shape),
# rocprofv3 --kernel-trace --stats of the DRIVER'S EXACT COMMAND and of the default command, the two PMC passes per launch shape turned into *_hbm_traffic.json
# (keyed on the library's build signature), and the shop-seeding ILP measurement with its SQ counters.   usage: tools/round5_artefacts.sh <tag>
set -u
tag="$1"; out="gpurun_out/$tag"; mkdir -p "$out"
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
(timeout 1500 python -m pytest tests -m gpu -q > "$out/gpu_tests.txt" 2>&1; echo rc=$? >> "$out/gpu_tests.txt"); tail -2 "$out/gpu_tests.txt"
# longer randomised parity runs than the suite holds (every record byte against the oracle; tools/stress_parity.py)
(for mode in "" "wide 5 6" "wide 11 6" long consumables; do echo "== mode [$mode] STRIDE=384"; STRIDE=384 timeout 900 python tools/stress_parity.py $mode 2>&1 | grep -v amdgpu.ids; done) > "$out/stress_parity.txt" 2>&1; grep -c "^ok" "$out/stress_parity.txt"; grep "STRESS OK" "$out/stress_parity.txt" | tr '\n' ' '; echo
python bench.py > "$out/bench_default.json" 2> "$out/bench_default.err"
python bench.py --gpus 1 --steps 20 --warmup 5 > "$out/bench_driver_shape.json" 2> "$out/bench_driver_shape.err"
rocprofv3 --kernel-trace --stats -d "$out/prof_driver" -o runc -- python3 bench.py --gpus 1 --steps 20 --warmup 5 > "$out/bench_driver_prof.json" 2> "$out/bench_driver_prof.err"
python tools/rocpd_summary.py "$(find $out/prof_driver -name '*.db' | head -1)" "$out/driver_cmd_kernel_stats.txt" > /dev/null 2>&1
rocprofv3 --kernel-trace --stats -d "$out/prof_default" -o runc -- python3 bench.py --no-cpu-baseline --no-step-path > "$out/bench_default_prof.json" 2> "$out/bench_default_prof.err"
python tools/rocpd_summary.py "$(find $out/prof_default -name '*.db' | head -1)" "$out/kernel_stats.txt" > /dev/null 2>&1
# a run made of 20-step launches only: what the launch of the driver's shape takes under rocprofv3 (median / steady mean per kernel)
rocprofv3 --kernel-trace --stats -d "$out/prof_t20" -o runc -- python3 bench.py --no-cpu-baseline --no-step-path --no-small-n --internal-warmup-launches 0 --chunk 20 --steps 2000 --warmup 400 --samples 0 > "$out/bench_t20_prof.json" 2> "$out/bench_t20_prof.err"
python tools/kernel_medians.py "$(find $out/prof_t20 -name '*.db' | head -1)" > "$out/t20_only_kernel_medians.txt" 2>&1
rm -rf "$out/prof_driver" "$out/prof_default" "$out/prof_t20"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$out/pmc_$c" -o runc -- python3 bench.py --no-cpu-baseline --no-step-path --no-small-n --steps 3720 --warmup 3720 --samples 0 > "$out/pmc_$c.json" 2> "$out/pmc_$c.err"
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$out/pmc20_$c" -o runc -- python3 bench.py --no-cpu-baseline --no-step-path --no-small-n --internal-warmup-launches 0 --chunk 20 --steps 2000 --warmup 400 --samples 0 > "$out/pmc20_$c.json" 2> "$out/pmc20_$c.err"
done
python tools/hbm_traffic.py "$out/pmc_FETCH_SIZE/runc_counter_collection.csv" "$out/pmc_WRITE_SIZE/runc_counter_collection.csv" 65536 372 "$out/hbm_traffic.json" "Round-5 build." > /dev/null 2>&1
python tools/hbm_traffic.py "$out/pmc20_FETCH_SIZE/runc_counter_collection.csv" "$out/pmc20_WRITE_SIZE/runc_counter_collection.csv" 65536 20 "$out/T20_hbm_traffic.json" "Round-5 build, the driver's launch shape (20 fused steps per launch)." > /dev/null 2>&1
for d in pmc_FETCH_SIZE pmc_WRITE_SIZE pmc20_FETCH_SIZE pmc20_WRITE_SIZE; do
  (head -1 "$out/$d/runc_counter_collection.csv"; grep "bg_engine" "$out/$d/runc_counter_collection.csv" | head -400) > "$out/$d.csv" 2>/dev/null
  rm -rf "$out/$d"
done
# the refill in pieces (the default) against the whole refill beside the next launches (BG_REFILL_SLICED=0), interleaved, at the driver's shape
for rep in 1 2 3; do for sl in 1 0; do
  BG_REFILL_SLICED=$sl python bench.py --no-cpu-baseline --no-step-path --no-small-n --steps 20 --warmup 5 --samples 60 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; s=d['samples']
print('BG_REFILL_SLICED=$sl rep $rep: value %.3f G  wall frac %.4f  kernel frac %.4f  sustained %.3f G  samples median %.3f  p10 %.3f  min %.3f  min/median %.3f' % (d['value']/1e9, r['frac'], r['kernel_frac'], d['sustained']['value']/1e9, s['median']/1e9, s['p10']/1e9, s['min']/1e9, s['min_over_median']))"
done; done > "$out/refill_pieces_ab.txt" 2>&1
[ -n "${SKIP_SHOP_ILP:-}" ] && exit 0
# shop seeding: one stream per lane against two interleaved (BG_SHOP_ILP), the kernel's SQ counters with nothing beside it (synchronous refill)
export BG_ASYNC_REFILL=0
for ilp in 1 2; do
  i=0
  for set in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU" "SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY"; do
    i=$((i+1))
    BG_SHOP_ILP=$ilp rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$out/shop_ilp${ilp}_$i" -o runc -- python3 bench.py --no-cpu-baseline --no-step-path --no-small-n --steps 1488 --warmup 744 --samples 0 --internal-warmup-launches 8 > /dev/null 2> "$out/shop_ilp${ilp}_$i.err"
  done
done
python - "$out" <<'PY' > "$out/shop_seeding_ilp.txt" 2>&1
import csv, glob, sys
out = sys.argv[1]
print("bg_refill_shop_kernel (one stream per lane) against bg_refill_shop2_kernel (two streams per lane, interleaved; BG_SHOP_ILP=2): mean per launch with NOTHING beside the kernel")
print("(BG_ASYNC_REFILL=0: the refill runs between the launches; rocprofv3 --kernel-trace --pmc <set>, one set per pass; durations from the kernel trace of the same passes)")
for ilp in (1, 2):
    acc, dur = {}, []
    for path in sorted(glob.glob(f"{out}/shop_ilp{ilp}_*/runc_counter_collection.csv")):
        for r in csv.DictReader(open(path)):
            if "bg_refill_shop" in r["Kernel_Name"]:
                acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    for path in sorted(glob.glob(f"{out}/shop_ilp{ilp}_*/runc_kernel_trace.csv")):
        for r in csv.DictReader(open(path)):
            if "bg_refill_shop" in r["Kernel_Name"]:
                dur.append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3)
    dur = dur[len(dur) // 4:]
    print(f"BG_SHOP_ILP={ilp}: kernel {sum(dur) / max(len(dur), 1):9.1f} us per launch ({len(dur)} launches)")
    for k, v in sorted(acc.items()):
        v = v[len(v) // 4:]
        print(f"    {k:28s} {sum(v) / len(v):18,.0f}")
PY
cat "$out/shop_seeding_ilp.txt"
rm -rf "$out"/shop_ilp*_[0-9]
unset BG_ASYNC_REFILL
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
head -12 "$out/driver_cmd_kernel_stats.txt"
