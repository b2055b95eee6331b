#!/bin/bash
# rocprofv3 kernel statistics (EVERY kernel, tagged by the library's profiler or not) of one python tool run on the GPU box.
# usage: tools/trace_stats.sh <tag> <tool.py> [args...]      -> gpurun_out/<tag>/stats.txt (top 16 kernels by total time)
set -eu
R="$(cd "$(dirname "$0")/.." && pwd)"
TAG="$1"; shift
O="$R/gpurun_out/$TAG"
rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$O" -- python3 "$R/$1" "${@:2}" > "$O/log" 2>&1
cd "$R"
python3 - "$O" <<'PY' | tee "$O/stats.txt"
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:16]:
    print("%-100s %6s %9.2f us %6s %%" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
find "$O" -name "*trace.csv" -delete
