#!/bin/bash
# Profiles the default bench run on the GPU box; usage: tools/profile_round.sh [tag]   (outputs in gpurun_out/<tag>)
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
ls -R "$O" | head -40
du -sh "$O"
