#!/bin/bash
out=gpurun_out/r05k; mkdir -p $out; export TMPDIR=/tmp
for cfg in "256 1 372" "4096 16 372"; do set -- $cfg
  BALATRO_MI355X_LIB=build/variants/e3t.so BG_E3_CFG=113 BG_E3_EPW=$2 N=$1 T=$3 WARM=$3 timeout 300 python tools/e3_timing.py 2>&1 | grep -v amdgpu.ids | tee $out/e3t_N$1_epw$2_T$3.txt
done
for T in 372 20; do BALATRO_MI355X_LIB=build/variants/e3t.so N=65536 T=$T WARM=$T timeout 300 python tools/e3_timing.py 2>&1 | grep -v amdgpu.ids | tee $out/e3t_N65536_T$T.txt; done
