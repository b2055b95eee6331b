#!/bin/bash
# usage: tools/bench_env.sh "VAR=1 VAR2=2" "VAR=3" ...   -> steps/s per environment
for e in "$@"; do
  v=$(env $e python bench.py --no-cpu-baseline --steps 192 --warmup 96 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['scoring_gemm']['achieved'])")
  echo "[$e] -> $v"
done
