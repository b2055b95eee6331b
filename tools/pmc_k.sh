#!/bin/bash
# SQ counters of the 16-wave split-bf16 kernel on one product: tools/pmc_k.sh <tag> LAYOUT M N K   -> gpurun_out/<tag>/summary.txt
set -u
R="$(cd "$(dirname "$0")/.." && pwd)"
TAG="$1"; shift
O="$R/gpurun_out/$TAG"
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
export GANMF_MFMA=f32 GANMF_X3KG=1
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u > "$O/sq_counters.txt"
for V in ${PMC_VARIANTS:-1 0}; do
  export GANMF_TUNE=kg=4,ring=3,rotate=$V
  i=0
  for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
             "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_LDS_UNALIGNED_STALL" \
             "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_VALU_MFMA_COEXEC_CYCLES SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL"; do
    i=$((i+1))
    rocprofv3 --kernel-trace --pmc $SET --output-format csv -d "$O/pmc_r${V}_$i" -- python3 "$R/tools/gemm_one.py" "$@" 64 1 5 > "$O/pmc_r${V}_$i.log" 2>&1 || true
    f=$(find "$O/pmc_r${V}_$i" -name "*counter_collection.csv" | head -1)
    echo "== rotate=$V set $i" >> "$O/summary.txt"
    python3 "$R/tools/pmc_summary.py" "$f" bf16k >> "$O/summary.txt" 2>&1 || tail -3 "$O/pmc_r${V}_$i.log" >> "$O/summary.txt"
  done
done
cat "$O/summary.txt"
find "$O" -name "*.csv" -size +2M -delete
