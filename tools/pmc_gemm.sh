#!/bin/bash
# SQ counters of one GEMM under the kernel variants; usage: tools/pmc_gemm.sh <tag> LAYOUT M N K  (outputs gpurun_out/<tag>)
set -eu
R="$(cd "$(dirname "$0")/.." && pwd)"
TAG="$1"; shift
O="$R/gpurun_out/$TAG"
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
for V in 0 1; do
  export GANMF_MFMA=f32 GANMF_TUNE=persist=$V
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT \
    --output-format csv -d "$O/pmc_p$V" -- python3 "$R/tools/gemm_one.py" "$@" 128 0 5 > "$O/pmc_p$V.log" 2>&1 || true
  f=$(find "$O/pmc_p$V" -name "*counter_collection.csv" | head -1)
  echo "== GANMF_TUNE=persist=$V" >> "$O/summary.txt"
  python3 "$R/tools/pmc_summary.py" "$f" gemm >> "$O/summary.txt" 2>&1 || true
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE FETCH_SIZE --output-format csv -d "$O/pmc_f$V" -- python3 "$R/tools/gemm_one.py" "$@" 128 0 5 > "$O/pmc_f$V.log" 2>&1 || true
  f=$(find "$O/pmc_f$V" -name "*counter_collection.csv" | head -1)
  python3 "$R/tools/pmc_summary.py" "$f" gemm >> "$O/summary.txt" 2>&1 || true
done
cat "$O/summary.txt"
find "$O" -name "*.csv" -size +2M -delete
