#!/bin/bash
# effective clock (GRBM_GUI_ACTIVE / 8 / duration) of the persistent scoring GEMM under GANMF_PERSIST_DIAG variants
set -eu
R="$(cd "$(dirname "$0")/.." && pwd)"
O="$R/gpurun_out/${1:-clock}"
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
export GANMF_MFMA=f32 GANMF_TUNE=persist=1
for D in 0 1 7; do
  export GANMF_PERSIST_DIAG=$D
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d "$O/d$D" -- python3 "$R/tools/gemm_one.py" NT 6040 3706 250 128 0 20 > "$O/d$D.log" 2>&1 || true
  f=$(find "$O/d$D" -name "*counter_collection.csv" | head -1)
  echo "== DIAG $D"; python3 "$R/tools/pmc_summary.py" "$f" gemm
done
find "$O" -name "*.csv" -size +1M -delete
