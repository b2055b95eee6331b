#!/bin/bash
# rocprofv3 kernel trace of one python tool + tools/timeline.py window.
# usage: tools/trace_cmd_timeline.sh <tag> <first> <count> <anchor> <tool.py> [args...]   -> gpurun_out/<tag>/timeline.txt
set -eu
R="$(cd "$(dirname "$0")/.." && pwd)"
TAG="$1"; FIRST="$2"; COUNT="$3"; ANCHOR="$4"; shift 4
O="$R/gpurun_out/$TAG"
rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$O" -- python3 "$R/$1" "${@:2}" > "$O/log" 2>&1
cd "$R"
python3 tools/timeline.py "$(ls $O/*/*kernel_trace.csv | head -1)" "$FIRST" "$COUNT" "$ANCHOR" > "$O/timeline.txt"
find "$O" -name "*trace.csv" -delete
