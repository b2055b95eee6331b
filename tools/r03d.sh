set -x
mkdir -p gpurun_out/r03d
python -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py tests/test_gpu_dist_local.py tests/test_gpu_sharded_fit.py -q -m gpu -s -k "sparse or rccl or sharded or bitwise or process" > gpurun_out/r03d/t.log 2>&1; echo "t rc=$?"; tail -5 gpurun_out/r03d/t.log
python tools/c1_bench.py > gpurun_out/r03d/c1.log 2>&1; cat gpurun_out/r03d/c1.log
python bench.py --no-cpu-baseline > gpurun_out/r03d/bench.json 2> gpurun_out/r03d/bench.err
GANMF_BENCH_FORCE_COMM=1 python bench.py --no-cpu-baseline > gpurun_out/r03d/bench_fc.json 2> gpurun_out/r03d/bench_fc.err
GANMF_LANE_EVENT_FENCE=1 GANMF_BENCH_FORCE_COMM=1 python bench.py --no-cpu-baseline > gpurun_out/r03d/bench_fc_fence.json 2> gpurun_out/r03d/bench_fc_fence.err
python -c "
import json
for f in ('bench','bench_fc','bench_fc_fence'):
    d=json.load(open('gpurun_out/r03d/%s.json'%f)); print(f, d['value'], d['roofline']['frac'], d['roofline'].get('frac_time_weighted'))
"
cd /tmp && export TMPDIR=/tmp
GANMF_BENCH_FORCE_COMM=1 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r03d/trace_fc -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 64 --warmup 32 > $GRAFT_REPO_ROOT/gpurun_out/r03d/trace_fc.log 2>&1
cd $GRAFT_REPO_ROOT
python3 tools/timeline.py "$(ls gpurun_out/r03d/trace_fc/*/*_kernel_trace.csv | head -1)" 40 2 > gpurun_out/r03d/timeline_fc.txt; cat gpurun_out/r03d/timeline_fc.txt
find gpurun_out/r03d -name "*_kernel_trace.csv" -delete
