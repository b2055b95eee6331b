mkdir -p gpurun_out/r03f
GANMF_TIME_EPOCH=1 GANMF_BENCH_FORCE_COMM=1 python bench.py --no-cpu-baseline --steps 94 --warmup 94 2>&1 >/dev/null | grep "ganmf epoch" | tail -4
GANMF_TIME_EPOCH=1 python bench.py --no-cpu-baseline --steps 94 --warmup 94 2>&1 >/dev/null | grep "ganmf epoch" | tail -4
