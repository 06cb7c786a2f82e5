#!/bin/bash
# classification by count masks, the hand codes with one branch, the empty-hand draw: parity, batch cycles before / after, interleaved A/B
out=gpurun_out/r05r; mkdir -p $out; export TMPDIR=/tmp
(timeout 900 python -m pytest tests/test_gpu_operators.py tests/test_gpu_parity.py -m gpu -x -q -k "classify or golden_trace or every_engine or full_size_slice or step_vs_oracle or consumables or immolate or fused_rollout_vs_oracle or forced or rare or sim_" > $out/gpu_tests.txt 2>&1; echo rc=$? >> $out/gpu_tests.txt); tail -3 $out/gpu_tests.txt
for v in e3t_base e3t; do for T in 20 372; do BALATRO_MI355X_LIB=build/variants/$v.so N=65536 T=$T WARM=$T timeout 300 python tools/e3_timing.py 2>&1 | grep -E "launch|SERVICE p|SERVICE o|OWNER \(" | sed "s/^/$v T=$T: /" | tee -a $out/e3t.txt; done; done
bash tools/ab_libs2.sh $out/ab 4 balatro_gym_amd/libbalatro_mi355x.so build/variants/base.so > $out/ab.txt 2>&1; cat $out/ab.txt
