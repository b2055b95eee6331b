#!/usr/bin/env python3
"""Best (tile, nsplit) per step GEMM shape under each K-loop arithmetic: python tools/gemm_modes.py
Runs each mode in its own process (GANMF_MFMA is read when the library plans a GEMM)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPES = [("NT", 128, 3706, 250), ("NN", 256, 992, 3707), ("NN", 256, 3706, 993), ("NT", 256, 992, 3706),
          ("NT", 128, 3706, 992), ("NN", 128, 250, 3706), ("NT", 128, 992, 3706)]
CODE = r"""
import sys; sys.path.insert(0, %r)
from tools.gemm_bench import run
for layout, M, N, K in %r:
    res = []
    for tile in (64, 128):
        for ns in (1, 2, 4, 8, 16, 32):
            if ns * 64 > K + 63: continue
            ms, tf = run(layout, M, N, K, tile, ns, 20)
            res.append((ms * 1e3, tile, ns))
    res.sort()
    print("%%s %%5dx%%5dx%%5d  " %% (layout, M, N, K) + "  ".join("%%.1fus(t%%d,s%%d)" %% r for r in res[:4]), flush=True)
"""
for mode in sys.argv[1:] or ["f32", "bf16x3", "bf16"]:
    print("== GANMF_MFMA=%s" % mode, flush=True)
    subprocess.run([sys.executable, "-c", CODE % (ROOT, SHAPES)], env=dict(os.environ, GANMF_MFMA=mode), check=False)
