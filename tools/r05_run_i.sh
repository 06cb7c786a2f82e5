#!/bin/bash
# single-env latency anatomy: one env per workgroup (BG_E3_EPW=1), probes inside the play path
out=gpurun_out/r05i; mkdir -p $out; export TMPDIR=/tmp
for cfg in "256 1" "1024 4" "4096 16"; do set -- $cfg
  BALATRO_MI355X_LIB=build/variants/pr.so BG_E3_CFG=113 BG_E3_EPW=$2 N=$1 T=372 WARM=372 timeout 300 python tools/probes4.py > $out/probes_N$1_epw$2.txt 2>&1; cat $out/probes_N$1_epw$2.txt | grep -v amdgpu.ids
done
