#!/bin/bash
# usage (DIAG build): tools/wgs_stamps.sh "0 1 6 7 ..."   -> workgroup timelines of wgrad_stream_kernel per GANMF_WGS_DIAG value
for d in $1; do
  GANMF_WGS_STAMPS=1 GANMF_WGS_DIAG=$d python bench.py --no-cpu-baseline --steps 40 --warmup 60 2>&1 >/dev/null | grep -A6 "wgs stamps"
done
