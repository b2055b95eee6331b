set -x
mkdir -p gpurun_out/r03c
python -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py tests/test_gpu_dist_local.py -q -m gpu -s -k "sparse or rccl or sharded or full_shape or bitwise" > gpurun_out/r03c/t.log 2>&1; echo "t rc=$?"; tail -5 gpurun_out/r03c/t.log
python tools/c1_bench.py > gpurun_out/r03c/c1.log 2>&1; cat gpurun_out/r03c/c1.log
cd /tmp && export TMPDIR=/tmp
GANMF_BENCH_FORCE_COMM=1 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r03c/trace_fc -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 64 --warmup 32 > $GRAFT_REPO_ROOT/gpurun_out/r03c/trace_fc.log 2>&1
cd $GRAFT_REPO_ROOT
python3 tools/timeline.py "$(ls gpurun_out/r03c/trace_fc/*/*_kernel_trace.csv | head -1)" 40 2 > gpurun_out/r03c/timeline_fc.txt; cat gpurun_out/r03c/timeline_fc.txt
find gpurun_out/r03c -name "*_kernel_trace.csv" -size +20M -delete
