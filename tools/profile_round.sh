#!/bin/bash
# Profiles the default bench run on the GPU box; usage: tools/profile_round.sh [tag]   (outputs in gpurun_out/<tag>)
#   bench.json             the bench line (default flags)
#   plan.log               [ganmf plan] lines: tile / ring / K groups / split / arithmetic per GEMM class
#   trace/                 rocprofv3 --kernel-trace --stats of `python3 bench.py --no-cpu-baseline`
#   pmc_fetch/ pmc_write/  FETCH_SIZE and WRITE_SIZE, one counter per pass (MI355X_MICROARCH: separate --pmc passes)
#   pmc_sq/                SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES (MFMA utilisation per class)
#   classes.md, classes_sq.md, traffic.json   the same, per kernel class by dispatch order (tools/step_classes.py)
set -eux
R="$(cd "$(dirname "$0")/.." && pwd)"
test -f "$R/bench.py"
TAG="${1:-r02}"
O="$R/gpurun_out/$TAG"
cd /tmp && export TMPDIR=/tmp
rm -rf "$O"; mkdir -p "$O"
cd "$R"
python bench.py > "$O"/bench.json 2> "$O"/bench.err
GANMF_DEBUG_PLAN=1 python bench.py --no-cpu-baseline 2>&1 >/dev/null | grep "ganmf plan" > "$O"/plan.log
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$O"/trace -- python3 "$R"/bench.py --no-cpu-baseline > "$O"/trace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$O"/pmc_fetch -- python3 "$R"/bench.py --no-cpu-baseline --steps 64 --warmup 16 > "$O"/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$O"/pmc_write -- python3 "$R"/bench.py --no-cpu-baseline --steps 64 --warmup 16 > "$O"/pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d "$O"/pmc_sq -- python3 "$R"/bench.py --no-cpu-baseline --steps 64 --warmup 16 > "$O"/pmc_sq.log 2>&1
cd "$R"
python3 tools/step_classes.py "$(ls "$O"/trace/*/*_kernel_trace.csv | head -1)" > "$O"/classes.md
python3 tools/step_classes.py "$(ls "$O"/pmc_sq/*/*_counter_collection.csv | head -1)" > "$O"/classes_sq.md
python3 tools/collect_traffic.py "$O"/pmc_fetch "$O"/pmc_write > "$O"/traffic.json
python3 tools/summarize_profile.py "$(ls "$O"/trace/*/*_kernel_trace.csv | head -1)" > "$O"/kernel_summary.md
# raw counter dumps are large: keep the per-class summaries, drop what exceeds the merge budget
find "$O" -name "*counter_collection.csv" -size +8M -delete
cat "$O"/classes.md
du -sh "$O"
