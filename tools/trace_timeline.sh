#!/bin/bash
# rocprofv3 kernel trace of a short bench.py run + tools/timeline.py window (one discriminator step, anchored on an encode launch).
# usage: tools/trace_timeline.sh <tag> [first=60] [count=2] [anchor]     (environment, e.g. GANMF_TUNE, is inherited)  -> gpurun_out/<tag>/timeline.txt
set -eu
R="$(cd "$(dirname "$0")/.." && pwd)"
TAG="$1"; FIRST="${2:-60}"; COUNT="${3:-2}"; ANCHOR="${4:-de_dcoef_kernel}"
O="$R/gpurun_out/$TAG"
rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$O" -- python3 "$R/bench.py" --no-cpu-baseline --steps 64 --warmup 32 > "$O/log" 2>&1
cd "$R"
python3 tools/timeline.py "$(ls $O/*/*kernel_trace.csv | head -1)" "$FIRST" "$COUNT" "$ANCHOR" > "$O/timeline.txt"
find "$O" -name "*trace.csv" -delete
