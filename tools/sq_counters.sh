#!/bin/bash
# GPU box: SQ / TA counters of the rollout kernel alone on the GPU (synchronous refill), one counter set per rocprofv3 pass.
# usage: tools/sq_counters.sh <tag> [372|20]   -> gpurun_out/<tag>/sq_<i>/..., summary by tools/sq_summary.py  (20: every launch of the pass fuses 20 steps, the driver's shape)
set -u
tag="$1"; shape="${2:-372}"; out="gpurun_out/$tag"; mkdir -p "$out"
if [ "$shape" = 20 ]; then args="--internal-warmup-launches 0 --chunk 20 --steps 2000 --warmup 400 --samples 0"; else args="--steps 1488 --warmup 744 --samples 0"; fi
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp BG_ASYNC_REFILL=0
i=0
for set in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES" "SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU" "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INSTS_SMEM" \
           "SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQ_IFETCH SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_WAVE32_LDS"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$out/sq_$i" -o runc -- python3 bench.py --no-cpu-baseline --no-step-path --no-small-n $args > "$out/sq_$i.json" 2> "$out/sq_$i.err"
done
ls $out/sq_*/runc_counter_collection.csv
