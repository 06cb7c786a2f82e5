#!/bin/bash
# GPU box: everything round 4 commits under profiles/ for the shipped library -- bench lines (default shape, the driver's shape, 4 096 envs),
# rocprofv3 kernel statistics of the default command, the two PMC passes per launch shape turned into *_hbm_traffic.json (keyed on the
# library's build signature), and the engine-by-engine comparison on this one box.   usage: tools/round4_artefacts.sh <tag>
set -u
tag="$1"; out="gpurun_out/$tag"
cd "$(dirname "$0")/.." || exit 1
bash tools/round_artefacts.sh "$tag" > /dev/null 2>&1
python tools/rocpd_summary.py "$out/prof/runc_results.db" "$out/kernel_stats.txt" > /dev/null 2>&1
python tools/hbm_traffic.py "$out/pmc_FETCH_SIZE/runc_counter_collection.csv" "$out/pmc_WRITE_SIZE/runc_counter_collection.csv" 65536 372 "$out/hbm_traffic.json" "Round-4 build: engine 3 (owner waves + service waves in one workgroup)." > /dev/null 2>&1
python tools/hbm_traffic.py "$out/pmc20_FETCH_SIZE/runc_counter_collection.csv" "$out/pmc20_WRITE_SIZE/runc_counter_collection.csv" 65536 20 "$out/T20_hbm_traffic.json" "Round-4 build, the driver's launch shape (20 fused steps per launch)." > /dev/null 2>&1
# the PMC passes themselves are kept small: the per-launch rows of the rollout kernel only
for d in pmc_FETCH_SIZE pmc_WRITE_SIZE pmc20_FETCH_SIZE pmc20_WRITE_SIZE; do
  (head -1 "$out/$d/runc_counter_collection.csv"; grep "bg_engine" "$out/$d/runc_counter_collection.csv" | head -400) > "$out/$d.csv" 2>/dev/null
  rm -rf "$out/$d"
done
rm -rf "$out/prof"
for e in 1 2 3; do for rep in 1 2; do
  BG_ENGINE=$e timeout 200 python bench.py --no-cpu-baseline --no-step-path --no-small-n > "$out/engine${e}_default_$rep.json" 2>/dev/null
  BG_ENGINE=$e timeout 200 python bench.py --no-cpu-baseline --no-step-path --no-small-n --steps 20 --warmup 5 > "$out/engine${e}_T20_$rep.json" 2>/dev/null
done; done
python - "$out" <<'PY'
import glob, json, os, sys
out = sys.argv[1]
lines = ["engine (1 = bg_engine.h workers + copiers, 2 = bg_engine2.h owner kernel + service kernel, 3 = bg_engine3.h owner + service waves in one workgroup),",
         "65 536 envs, BASELINE configs[2], one box, interleaved: file, G env-steps/s (value), roofline.frac of 8 TB/s (kernel), sustained G env-steps/s, mean launch us"]
for f in sorted(glob.glob(os.path.join(out, "engine*_*.json"))):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        lines.append(f"{os.path.basename(f):28s} {d['value'] / 1e9:6.3f} {d['roofline']['frac']:.4f} {d['sustained']['value'] / 1e9:6.3f} {d['roofline']['mean_launch_us']:8.1f}")
    except Exception as ex:
        lines.append(f"{os.path.basename(f)} failed: {ex}")
open(os.path.join(out, "engines_ab.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
for name in ("bench_default", "bench_driver_shape", "bench_4096", "hbm_traffic", "T20_hbm_traffic"):
    try:
        d = json.loads(open(os.path.join(out, name + ".json")).read().strip().splitlines()[-1]) if name.startswith("bench") else json.load(open(os.path.join(out, name + ".json")))
        if name.startswith("bench"):
            print(name, round(d["value"] / 1e9, 3), round(d["roofline"]["frac"], 4), d["roofline"]["traffic"], d.get("small_n", {}).get("value"))
        else:
            print(name, d["hbm_bytes_per_env_step"], d["device_code_sha"])
    except Exception as ex:
        print(name, "failed:", ex)
PY
cat "$out/kernel_stats.txt" 2>/dev/null | head -12
