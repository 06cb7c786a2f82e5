#!/bin/bash
# de-branched bg_chain_main: parity subset, interleaved A/B against the library without it, single-env probes before / after
out=gpurun_out/r05j; mkdir -p $out; export TMPDIR=/tmp
(timeout 900 python -m pytest tests/test_gpu_operators.py tests/test_gpu_parity.py -m gpu -x -q -k "score_hand or golden_trace or repeated_jokers or every_engine or full_size_slice or step_vs_oracle or global_stream" > $out/gpu_tests.txt 2>&1; echo rc=$? >> $out/gpu_tests.txt); tail -3 $out/gpu_tests.txt
bash tools/ab_libs2.sh $out/ab 3 balatro_gym_amd/libbalatro_mi355x.so build/variants/base.so > $out/ab.txt 2>&1; cat $out/ab.txt
for v in pr pr2; do BALATRO_MI355X_LIB=build/variants/$v.so BG_E3_CFG=113 BG_E3_EPW=1 N=256 T=372 WARM=372 timeout 300 python tools/probes4.py 2>&1 | grep -v amdgpu.ids > $out/probes_single_$v.txt; grep -E "launch|probe 14|probe 15|probe 20|probe 26|probe 22" $out/probes_single_$v.txt; done
