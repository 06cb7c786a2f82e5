#!/bin/bash
out=gpurun_out/r05m; mkdir -p $out; export TMPDIR=/tmp
(timeout 900 python -m pytest tests/test_gpu_operators.py tests/test_gpu_parity.py -m gpu -x -q -k "golden_trace or score_hand or shop_stream or every_engine or full_size_slice or step_vs_oracle or consumables_rollout or fused_rollout_vs_oracle or repeated_jokers or global_stream or curriculum" > $out/gpu_tests.txt 2>&1; echo rc=$? >> $out/gpu_tests.txt); tail -3 $out/gpu_tests.txt
bash tools/ab_libs2.sh $out/ab 3 balatro_gym_amd/libbalatro_mi355x.so build/variants/noblood.so build/variants/base.so > $out/ab.txt 2>&1; cat $out/ab.txt
