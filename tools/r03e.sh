mkdir -p gpurun_out/r03e
python -m pytest tests/test_gpu_parity.py tests/test_gpu_dist_local.py tests/test_gpu_sharded_fit.py -q -m gpu -s -k "rccl or sharded or process" > gpurun_out/r03e/t.log 2>&1; echo "t rc=$?"; tail -3 gpurun_out/r03e/t.log
python bench.py --no-cpu-baseline > gpurun_out/r03e/bench.json 2> gpurun_out/r03e/bench.err
GANMF_BENCH_FORCE_COMM=1 python bench.py --no-cpu-baseline > gpurun_out/r03e/bench_fc.json 2> gpurun_out/r03e/bench_fc.err
GANMF_FORK_ATTACH=0 GANMF_BENCH_FORCE_COMM=1 python bench.py --no-cpu-baseline > gpurun_out/r03e/bench_fc_noattach.json 2> gpurun_out/r03e/bench_fc_noattach.err
python bench.py --no-cpu-baseline > gpurun_out/r03e/bench2.json 2> gpurun_out/r03e/bench2.err
python -c "
import json
for f in ('bench','bench_fc','bench_fc_noattach','bench2'):
    d=json.load(open('gpurun_out/r03e/%s.json'%f)); print(f, d['value'], d['roofline']['frac'], d['roofline'].get('frac_time_weighted'))
"
cd /tmp && export TMPDIR=/tmp
GANMF_BENCH_FORCE_COMM=1 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r03e/trace_fc -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 64 --warmup 32 > $GRAFT_REPO_ROOT/gpurun_out/r03e/trace_fc.log 2>&1
cd $GRAFT_REPO_ROOT
python3 tools/timeline.py "$(ls gpurun_out/r03e/trace_fc/*/*_kernel_trace.csv | head -1)" 40 1 > gpurun_out/r03e/timeline_fc_D.txt; cat gpurun_out/r03e/timeline_fc_D.txt
python3 tools/timeline.py "$(ls gpurun_out/r03e/trace_fc/*/*_kernel_trace.csv | head -1)" 70 1 > gpurun_out/r03e/timeline_fc_G.txt; cat gpurun_out/r03e/timeline_fc_G.txt
find gpurun_out/r03e -name "*_kernel_trace.csv" -delete
