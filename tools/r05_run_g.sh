#!/bin/bash
# small jobs: live envs per 64-env workgroup (BG_E3_EPW) at 4 096 / 8 192 / 16 384 envs, and the shapes' parity test
out=gpurun_out/r05g; mkdir -p $out; export TMPDIR=/tmp
(timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "every_engine or every_workgroup_shape or fused_rollout_vs_oracle or many_short" > $out/gpu_tests.txt 2>&1; echo rc=$? >> $out/gpu_tests.txt); tail -3 $out/gpu_tests.txt
for rep in 1 2; do
for epw in 64 32 16 8 0; do BG_E3_CFG=113 BG_E3_EPW=$epw NS=4096,8192,16384 timeout 300 python tools/small_n.py 2>/dev/null | sed "s/^/epw=$epw /"; done
BG_E3_CFG=213 NS=16384,32768 timeout 300 python tools/small_n.py 2>/dev/null | sed "s/^/cfg213 /"
BG_E3_CFG=113 BG_E3_EPW=64 NS=32768 timeout 300 python tools/small_n.py 2>/dev/null | sed "s/^/epw=64 /"
BG_E3_CFG=413 NS=32768 timeout 300 python tools/small_n.py 2>/dev/null | sed "s/^/cfg413 /"
done | tee $out/small_n.txt
