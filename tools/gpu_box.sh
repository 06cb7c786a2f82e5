#!/bin/bash
# GPU box: ONE parameterised script for what a round runs there (it replaces the 33 one-off tools/r05_run_*.sh of round 5).
#   usage: tools/gpu_box.sh <outdir under gpurun_out/> <step> [<step> ...]
#   steps:
#     micro            tools/micro/ldsdma (LDS-DMA semantics the engine's prefetches rely on) and tools/micro/litmus_peer (peer-written gather buffers)
#     tests            the whole GPU suite (-m gpu)
#     test:<expr>      pytest -m gpu -k <expr>
#     oldlib:<expr>    the same selection against build/variants/r05_shipped.so (a test that pins a fixed bug must FAIL there)
#     steppath:<reps>:<lib>[,<lib>...]   bench.py's step_path block (bg_step / bg_step_rows / bg_step_many) of several libraries, interleaved
#     ab:<reps>:<lib>[,<lib>...]   interleaved kernel-only bench (both launch shapes) of several libraries on this box; "tree" = the in-tree library
#     abenv:<reps>:<VAR=a,b,..>    the same for values of one environment variable of the in-tree library
#     timing:<lib>     tools/e3_timing.py with a -DBG_E3_TIMING library (cycles per batch / owner iteration), both shapes
#     probes:<lib>     tools/probes4.py with a -DBG_TIMING library (cycle probes of the play path at full load)
#     stress           tools/stress_parity.py in its five modes (every record byte against the oracle)
#     campaign:<k>     k seed offsets x the stress modes (one launch / many short launches) + the GPU suite under BG_TEST_SEED_OFFSET
#     sq               SQ counters of bg_engine3_kernel at both launch shapes (tools/sq_counters.sh passes)
#     final            the round's artefacts: bench lines, rocprofv3 --kernel-trace --stats of the driver's command and the default one, PMC traffic
set -u
cd "$(dirname "$0")/.." || exit 1
tag="$1"; shift
out="gpurun_out/$tag"; mkdir -p "$out"
export TMPDIR=/tmp
brief() {  # one line per bench JSON
  python - "$@" <<'PY'
import json, sys
for f in sys.argv[1:]:
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); r = d["roofline"]; s = d.get("samples", {})
        print("%-44s value %6.3f G  wall frac %.4f  kernel frac %.4f  launch %7.1f us  sustained %6.3f G  samples median %6.3f  min/med %.3f" % (
            f.split("/")[-1], d["value"] / 1e9, r["frac"], r.get("kernel_frac", 0), r["mean_launch_us"], d["sustained"]["value"] / 1e9,
            s.get("median", 0) / 1e9, s.get("min_over_median", 0)))
    except Exception as ex:
        print(f.split("/")[-1], "failed:", ex)
PY
}
libpath() { [ "$1" = tree ] && echo balatro_gym_amd/libbalatro_mi355x.so || { [ -f "$1" ] && echo "$1" || echo "build/variants/$1.so"; }; }
for step in "$@"; do
  case "$step" in
    micro)
      (timeout 120 tools/micro/ldsdma; echo rc=$?) > "$out/ldsdma.txt" 2>&1; tail -2 "$out/ldsdma.txt"
      (timeout 300 tools/micro/litmus_peer 100 65536; echo rc=$?) > "$out/litmus_peer.txt" 2>&1; cat "$out/litmus_peer.txt" ;;
    tests)
      (timeout 2400 python -m pytest tests -m gpu -x -q > "$out/gpu_tests.txt" 2>&1; echo rc=$? >> "$out/gpu_tests.txt"); tail -5 "$out/gpu_tests.txt" ;;
    test:*)
      (timeout 1200 python -m pytest tests -m gpu -x -q -k "${step#test:}" > "$out/test_sel.txt" 2>&1; echo rc=$? >> "$out/test_sel.txt"); tail -5 "$out/test_sel.txt" ;;
    oldlib:*)
      (BALATRO_MI355X_LIB=build/variants/r05_shipped.so timeout 1200 python -m pytest tests -m gpu -x -q -k "${step#oldlib:}" > "$out/test_oldlib.txt" 2>&1; echo rc=$? >> "$out/test_oldlib.txt"); tail -8 "$out/test_oldlib.txt" ;;
    ab:*)
      IFS=: read -r _ reps libs <<< "$step"
      for rep in $(seq 1 "$reps"); do for lib in ${libs//,/ }; do p=$(libpath "$lib"); name=$(basename "$p" .so); [ "$lib" = tree ] && name=tree
        BALATRO_MI355X_LIB=$p timeout 300 python bench.py --no-cpu-baseline --no-step-path --no-small-n --steps 20 --warmup 5 > "$out/T20_${name}_$rep.json" 2>/dev/null
        BALATRO_MI355X_LIB=$p timeout 300 python bench.py --no-cpu-baseline --no-step-path --no-small-n > "$out/default_${name}_$rep.json" 2>/dev/null
      done; done
      brief "$out"/T20_*.json "$out"/default_*.json | tee "$out/ab_summary.txt" ;;
    steppath:*)   # steppath:<reps>:<lib>[,<lib>...]: the bg_step / bg_step_rows / bg_step_many figures of bench.py's `step_path` block, interleaved
      IFS=: read -r _ reps libs <<< "$step"
      (for rep in $(seq 1 "$reps"); do for lib in ${libs//,/ }; do p=$(libpath "$lib"); name=$(basename "$p" .so); [ "$lib" = tree ] && name=tree
        BALATRO_MI355X_LIB=$p timeout 600 python bench.py --no-cpu-baseline --no-small-n --samples 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['step_path']
print('$name rep $rep:', '  '.join('%s %.3f G (kernel %.2f us / step, wall %.2f)' % (k, s[k]['value']/1e9, s[k]['kernel_ms_per_step']*1e3, s[k]['ms_per_step']*1e3) for k in ('bg_step','bg_step_rows','bg_step_many','bg_step_many_kept') if k in s), ' headline %.3f G' % (d['value']/1e9))"
      done; done) > "$out/step_path_ab.txt" 2>&1; cat "$out/step_path_ab.txt" ;;
    abenv:*)
      IFS=: read -r _ reps spec <<< "$step"; var=${spec%%=*}; vals=${spec#*=}
      for rep in $(seq 1 "$reps"); do for v in ${vals//,/ }; do
        env "$var=$v" timeout 300 python bench.py --no-cpu-baseline --no-step-path --no-small-n --steps 20 --warmup 5 > "$out/T20_${var}_${v}_$rep.json" 2>/dev/null
        env "$var=$v" timeout 300 python bench.py --no-cpu-baseline --no-step-path --no-small-n > "$out/default_${var}_${v}_$rep.json" 2>/dev/null
      done; done
      brief "$out"/T20_${var}_*.json "$out"/default_${var}_*.json | tee "$out/abenv_${var}_summary.txt" ;;
    timing:*)
      p=$(libpath "${step#timing:}")
      (for T in 372 20; do echo "== T $T"; BALATRO_MI355X_LIB=$p T=$T timeout 600 python tools/e3_timing.py 2>&1 | grep -v amdgpu.ids; done) > "$out/e3_timing_$(basename "$p" .so).txt" 2>&1; cat "$out/e3_timing_$(basename "$p" .so).txt" ;;
    probes:*)
      p=$(libpath "${step#probes:}")
      (BALATRO_MI355X_LIB=$p timeout 600 python tools/probes4.py) > "$out/probes_$(basename "$p" .so).txt" 2>&1; tail -70 "$out/probes_$(basename "$p" .so).txt" ;;
    stress)
      (for mode in "" "wide 5 6" "wide 11 6" long consumables; do echo "== mode [$mode] STRIDE=384"; STRIDE=384 timeout 900 python tools/stress_parity.py $mode 2>&1 | grep -v amdgpu.ids; done) > "$out/stress_parity.txt" 2>&1
      grep -c "^ok" "$out/stress_parity.txt"; grep "STRESS OK\|FAIL\|Error" "$out/stress_parity.txt" | tr '\n' ' '; echo ;;
    campaign:*)   # campaign:<k>: k seed offsets x (the five stress modes, once as ONE launch and once as short launches with the refill in pieces) + the GPU suite under BG_TEST_SEED_OFFSET
      k="${step#campaign:}"
      (python -c "from balatro_gym_amd import build; print('library', build.library_signature(), 'sources', build.source_signature())"
       for off in $(seq 1 "$k"); do for chunks in "" "20,13,30,7"; do for mode in "" "wide 5 6" long consumables; do
          echo "== SEED_OFFSET=$((off * 100003)) CHUNKS=[$chunks] mode [$mode]"; SEED_OFFSET=$((off * 100003)) CHUNKS=$chunks STRIDE=384 timeout 900 python tools/stress_parity.py $mode 2>&1 | grep -v amdgpu.ids
        done; done
        echo "== BG_TEST_SEED_OFFSET=$((off * 7919)): the GPU suite"; BG_TEST_SEED_OFFSET=$((off * 7919)) timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -2
      done) > "$out/stress_campaign.txt" 2>&1
      grep -c "^ok" "$out/stress_campaign.txt"; grep -c "STRESS OK" "$out/stress_campaign.txt"; grep "passed\|failed\|Error\|FAIL" "$out/stress_campaign.txt" | tr '\n' ' '; echo ;;
    sq)
      tools/sq_counters.sh "$tag/sq372" 372 > /dev/null 2>&1; python tools/sq_summary.py "$tag/sq372" bg_engine3_kernel > "$out/sq_counters.txt" 2>&1
      tools/sq_counters.sh "$tag/sq20" 20 > /dev/null 2>&1; python tools/sq_summary.py "$tag/sq20" bg_engine3_kernel > "$out/sq_counters_T20.txt" 2>&1
      rm -rf "$out"/sq372/sq_[0-9]* "$out"/sq20/sq_[0-9]*; head -40 "$out/sq_counters.txt" "$out/sq_counters_T20.txt" ;;
    final)
      tools/round_artefacts.sh "$tag" ;;
    *) echo "unknown step $step" ;;
  esac
done
