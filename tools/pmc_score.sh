#!/bin/bash
# SQ counters of the scoring product's kernels (tools/score_one.py); usage: tools/pmc_score.sh <tag>   -> gpurun_out/<tag>/summary.txt
set -eu
R="$(cd "$(dirname "$0")/.." && pwd)"
O="$R/gpurun_out/$1"
rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT \
  --output-format csv -d "$O/p1" -- python3 "$R/tools/score_one.py" 10 > "$O/p1.log" 2>&1 || true
python3 "$R/tools/pmc_summary.py" "$(find "$O/p1" -name "*counter_collection.csv" | head -1)" bf16p >> "$O/summary.txt" 2>&1 || true
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_VMEM SQ_INST_CYCLES_VMEM \
  --output-format csv -d "$O/p2" -- python3 "$R/tools/score_one.py" 10 > "$O/p2.log" 2>&1 || true
python3 "$R/tools/pmc_summary.py" "$(find "$O/p2" -name "*counter_collection.csv" | head -1)" bf16p >> "$O/summary.txt" 2>&1 || true
cat "$O/summary.txt"; tail -3 "$O/p2.log"
find "$O" -name "*.csv" -size +2M -delete
