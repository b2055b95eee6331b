#!/usr/bin/env python3
"""Runs ONE GEMM shape a few times (for rocprofv3 --pmc runs).  usage: gemm_one.py LAYOUT M N K [tile] [nsplit] [iters]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.gemm_bench import run  # noqa: E402

a = sys.argv[1:]
ms, tf = run(a[0], int(a[1]), int(a[2]), int(a[3]), int(a[4]) if len(a) > 4 else 0, int(a[5]) if len(a) > 5 else 0,
             int(a[6]) if len(a) > 6 else 5)
print("%s %s: %.2f us %.1f TF/s" % (a[0], "x".join(a[1:4]), ms * 1e3, tf))
