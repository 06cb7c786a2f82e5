#!/bin/bash
out=gpurun_out/r05d; mkdir -p $out; export TMPDIR=/tmp
bash tools/ab_libs2.sh $out/ab 2 balatro_gym_amd/libbalatro_mi355x.so build/variants/onepass.so > $out/ab.txt 2>&1; cat $out/ab.txt
for T in 20 372; do
  BALATRO_MI355X_LIB=build/variants/e3t.so T=$T WARM=$T timeout 300 python tools/e3_timing.py > $out/e3_timing_T$T.txt 2>&1; cat $out/e3_timing_T$T.txt
  BALATRO_MI355X_LIB=build/variants/pr.so T=$T WARM=$T timeout 300 python tools/probes4.py > $out/probes_T$T.txt 2>&1; cat $out/probes_T$T.txt
done
