#!/bin/bash
# probes of the play path at FULL load (what a batch pays per phase) beside one env per workgroup
out=gpurun_out/r05u; mkdir -p $out; export TMPDIR=/tmp
for cfg in "65536 372 0" "65536 20 0" "256 372 1"; do set -- $cfg
  BG_E3_EPW=$3 BALATRO_MI355X_LIB=build/variants/pr.so N=$1 T=$2 WARM=372 timeout 300 python tools/probes4.py 2>&1 | grep -v amdgpu.ids | tee -a $out/probes.txt
done
