#!/usr/bin/env python3
"""K sweep: separates the per-workgroup fixed cost from the per-K-tile cost."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.gemm_bench import run
for layout, M, N in (("TN", 993, 3706), ("NT", 993, 3706), ("NN", 256, 3706), ("NT", 6040, 3706)):
    for tile in (128, 64):
        row = []
        for K in (64, 128, 256, 512, 1024, 2048):
            ms, tf = run(layout, M, N, K, tile, 1, 20)
            row.append("K=%d: %.1fus" % (K, ms * 1e3))
        print(layout, M, N, "tile", tile, " | ".join(row), flush=True)
