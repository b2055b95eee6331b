#!/usr/bin/env python3
"""Split-K sweep of the K-heavy step GEMMs through ganmf_gemm_f32 (GEMM + reduce, warm caches): which split the cost
model should pick.  usage: python tools/split_sweep.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gemm_bench import run  # noqa: E402

CASES = [("encode D/G", "NN", 256, 992, 3707, [2, 3, 4, 5, 6, 8]), ("dE D", "NT", 256, 992, 3706, [2, 3, 4, 5, 6, 8]),
         ("decode G", "NN", 128, 3706, 993, [1, 2, 3, 4]), ("dE G", "NT", 128, 992, 3706, [4, 5, 6, 7, 8, 10]),
         ("dF G", "NT", 128, 3706, 992, [1, 2, 3, 4])]
for name, layout, M, N, K, splits in CASES:
    for ns in splits:
        ms, tf = run(layout, M, N, K, 64, ns, iters=50)
        print("%-10s %s %4dx%4dx%4d nsplit=%2d : %7.2f us" % (name, layout, M, N, K, ns, ms * 1e3), flush=True)
