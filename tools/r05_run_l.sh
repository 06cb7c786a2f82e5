#!/bin/bash
# SWAR shop inventory: parity subset, A/B against HEAD, timing
out=gpurun_out/r05l; mkdir -p $out; export TMPDIR=/tmp
(timeout 900 python -m pytest tests/test_gpu_operators.py tests/test_gpu_parity.py -m gpu -x -q -k "golden_trace or shop_stream or every_engine or full_size_slice or step_vs_oracle or consumables_rollout or fused_rollout_vs_oracle or rollout_properties" > $out/gpu_tests.txt 2>&1; echo rc=$? >> $out/gpu_tests.txt); tail -3 $out/gpu_tests.txt
bash tools/ab_libs2.sh $out/ab 3 balatro_gym_amd/libbalatro_mi355x.so build/variants/base.so > $out/ab.txt 2>&1; cat $out/ab.txt
for T in 372 20; do BALATRO_MI355X_LIB=build/variants/e3t.so N=65536 T=$T WARM=$T timeout 300 python tools/e3_timing.py 2>&1 | grep -E "launch|SERVICE p|SERVICE o|OWNER \(" | tee $out/e3t_N65536_T$T.txt; done
BALATRO_MI355X_LIB=build/variants/e3t.so BG_E3_CFG=113 BG_E3_EPW=1 N=256 T=372 WARM=372 timeout 300 python tools/e3_timing.py 2>&1 | grep -E "launch|SERVICE p|SERVICE o" | tee $out/e3t_single.txt
