#!/usr/bin/env python3
"""GPU micro-benchmark of the fp32 MFMA GEMM through the C ABI (ganmf_gemm_f32).
usage: python tools/gemm_bench.py [shape-set]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ganmf_amd.engine import gemm_f32  # noqa: E402

PEAK = 157.3
SETS = {
    "big": [("NT", 4096, 4096, 4096), ("NN", 4096, 4096, 4096), ("TN", 4096, 4096, 4096)],
    "step": [("NT", 128, 3706, 250), ("NN", 256, 992, 3707), ("NN", 256, 3706, 993), ("NT", 256, 992, 3706),
             ("TN", 993, 3706, 256), ("TN", 3707, 992, 256), ("NT", 128, 3706, 992), ("NN", 128, 250, 3706),
             ("TN", 3706, 250, 128), ("NT", 6040, 3706, 250)],
    # the K-heavy GEMMs of the C2 step whose outputs are too small to fill the chip without split-K
    "skinny": [("NN", 256, 992, 3707), ("NT", 256, 992, 3706), ("NT", 128, 992, 3706), ("NN", 128, 250, 3706)],
}


def run(layout, M, N, K, tile=0, nsplit=0, iters=20):
    akm, bkm = {"NT": (False, False), "NN": (False, True), "TN": (True, True)}[layout]
    rng = np.random.RandomState(0)
    A = rng.standard_normal((K, M) if akm else (M, K)).astype(np.float32)
    B = rng.standard_normal((K, N) if bkm else (N, K)).astype(np.float32)
    _, ms = gemm_f32(A, B, akm, bkm, tile=tile, nsplit=nsplit, iters=iters)
    tf = 2.0 * M * N * K / ms / 1e9
    return ms, tf


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "step"
    tiles = [int(x) for x in os.environ.get("TILES", "0").split(",")]
    splits = [int(x) for x in os.environ.get("SPLITS", "0").split(",")]
    for layout, M, N, K in SETS[which]:
        for tile in tiles:
            for ns in splits:
                ms, tf = run(layout, M, N, K, tile, ns)
                print("%s %5dx%5dx%5d tile=%3d nsplit=%2d ring=%s : %8.2f us  %6.1f TF/s (%4.1f%%)" % (
                    layout, M, N, K, tile, ns, os.environ.get("GANMF_TUNE", "auto"), ms * 1e3, tf, 100 * tf / PEAK), flush=True)
