#!/bin/bash
# second part of the randomised parity campaign: other games (SEED_OFFSET), mid-size launches (the refill in several pieces per launch), card states + consumables as short launches
out=gpurun_out/r05al; mkdir -p $out; export TMPDIR=/tmp
(
for sd in 41 42 43; do echo "== wide $sd 8, SEED_OFFSET=$((sd * 1000)), dense records"; SEED_OFFSET=$((sd * 1000)) timeout 900 python tools/stress_parity.py wide $sd 8 2>&1 | grep -v amdgpu.ids; done
for sd in 51 52; do echo "== wide $sd 6 as mid-size launches (CHUNKS=100,180,60,48), STRIDE=384, SEED_OFFSET=$((sd * 1000))"; SEED_OFFSET=$((sd * 1000)) STRIDE=384 CHUNKS=100,180,60,48 timeout 900 python tools/stress_parity.py wide $sd 6 2>&1 | grep -v amdgpu.ids; done
echo "== consumables (4 096 envs, all 52 ids, card states) as 20-step launches, STRIDE=384, SEED_OFFSET=7000"; SEED_OFFSET=7000 STRIDE=384 CHUNKS=20 timeout 1200 python tools/stress_parity.py consumables 2>&1 | grep -v amdgpu.ids
echo "== long (2 048 x 1 500, 1 024 x 1 200 with card states) as 48-step launches, SEED_OFFSET=9000"; SEED_OFFSET=9000 CHUNKS=48 timeout 1200 python tools/stress_parity.py long 2>&1 | grep -v amdgpu.ids
) > $out/stress_campaign2.txt 2>&1
grep -c "^ok" $out/stress_campaign2.txt; grep "STRESS OK" $out/stress_campaign2.txt | tr '\n' ' '; echo; grep -i -E "error|assert|Traceback" $out/stress_campaign2.txt | head -5
