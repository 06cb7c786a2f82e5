#!/bin/bash
out=gpurun_out/r05q; mkdir -p $out; export TMPDIR=/tmp
(BALATRO_MI355X_LIB=build/variants/kp.so timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "golden_trace or every_engine or full_size_slice or fused_rollout_vs_oracle or card_states or consumables_rollout" > $out/gpu_tests_kp.txt 2>&1; echo rc=$? >> $out/gpu_tests_kp.txt); tail -3 $out/gpu_tests_kp.txt
bash tools/ab_libs2.sh $out/ab 3 balatro_gym_amd/libbalatro_mi355x.so build/variants/kp.so > $out/ab.txt 2>&1; cat $out/ab.txt
