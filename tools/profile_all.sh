#!/bin/bash
# Whole evidence set of a round in one gpurun call: tools/profile_all.sh <tag>  ->  gpurun_out/<tag>/ (default bench profile, per-config
# tables and rocprofv3 statistics, LastFM sparse-path numbers, the data-parallel step's bench line and timeline)
set -x
T=${1:-r03}
bash tools/profile_round.sh $T > gpurun_out/${T}_profile_round.log 2>&1; echo "profile_round rc=$?"
python tools/config_profiles.py > gpurun_out/$T/config_profiles.md 2> gpurun_out/$T/config_profiles.err; echo "config_profiles rc=$?"; tail -2 gpurun_out/$T/config_profiles.err
for c in c1_defaults c1_tuned c3 c4_e32 c5 c5_f16; do
  bash tools/trace_stats.sh $T/trace_$c tools/config_profiles.py $c > /dev/null 2>&1; echo "trace $c rc=$?"
done
python tools/c1_bench.py > gpurun_out/$T/c1.log 2>&1
GANMF_BENCH_FORCE_COMM=1 python bench.py --no-cpu-baseline > gpurun_out/$T/bench_fc.json 2> gpurun_out/$T/bench_fc.err
cd /tmp && export TMPDIR=/tmp
GANMF_BENCH_FORCE_COMM=1 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$T/trace_fc -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 64 --warmup 32 > $GRAFT_REPO_ROOT/gpurun_out/$T/trace_fc.log 2>&1
cd $GRAFT_REPO_ROOT
python3 tools/timeline.py "$(ls gpurun_out/$T/trace_fc/*/*_kernel_trace.csv | head -1)" 40 1 de_dcoef_kernel > gpurun_out/$T/timeline_fc_D.txt
python3 tools/timeline.py "$(ls gpurun_out/$T/trace_fc/*/*_kernel_trace.csv | head -1)" 40 1 > gpurun_out/$T/timeline_fc_G.txt
find gpurun_out/$T -name "*_kernel_trace.csv" -delete
du -sh gpurun_out/$T
