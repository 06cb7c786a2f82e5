#!/bin/bash
# a longer randomised parity campaign than the artefacts run holds: every record byte against the oracle (tools/stress_parity.py), incl. the games as many SHORT launches
out=gpurun_out/r05ae; mkdir -p $out; export TMPDIR=/tmp
(
echo "== short (CHUNKS cycles of 20 / 13 / 30 / 7 / 20 / 20 and plain 20-step launches: the refill in pieces), STRIDE=384"; STRIDE=384 timeout 1500 python tools/stress_parity.py short 2>&1 | grep -v amdgpu.ids
for sd in 21 22 23 24; do echo "== wide $sd 8, SEED_OFFSET=$((sd * 1000)), STRIDE=384"; SEED_OFFSET=$((sd * 1000)) STRIDE=384 timeout 900 python tools/stress_parity.py wide $sd 8 2>&1 | grep -v amdgpu.ids; done
for sd in 31 32; do echo "== wide $sd 6 as short launches (CHUNKS=20,13,30), dense records"; CHUNKS=20,13,30 timeout 900 python tools/stress_parity.py wide $sd 6 2>&1 | grep -v amdgpu.ids; done
echo "== big (8 192 envs), BG_REFILL_SLICED=0 and 1 give the same bytes as the oracle"; BG_REFILL_SLICED=0 STRIDE=384 timeout 900 python tools/stress_parity.py big 2>&1 | grep -v amdgpu.ids
) > $out/stress_campaign.txt 2>&1
grep -c "^ok" $out/stress_campaign.txt; grep "STRESS OK" $out/stress_campaign.txt | tr '\n' ' '; echo; grep -i -E "error|assert|Traceback" $out/stress_campaign.txt | head -5
