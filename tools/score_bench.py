"""Scoring GEMM 6040 x 3706 x 250 (and 4096^3) through ganmf_gemm_f32 under the K-loop modes / kernel variants.
usage: python tools/score_bench.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ganmf_amd.engine import gemm_f32  # noqa: E402


def run(M, N, K, env, iters=30):
    for k, v in env.items():
        os.environ[k] = v
    rng = np.random.RandomState(0)
    A = rng.standard_normal((M, K)).astype(np.float32)
    B = rng.standard_normal((N, K)).astype(np.float32)
    best = 1e9
    for _ in range(3):
        _, ms = gemm_f32(A, B, False, False, tile=128, iters=iters)
        best = min(best, ms)
    return best


if __name__ == "__main__":
    for (M, N, K) in [(6040, 3706, 250), (4096, 4096, 4096), (25000, 50000 // 8, 250)]:
        for name, env in [("f32 one-tile", {"GANMF_MFMA": "f32", "GANMF_TUNE": "persist=0"}),
                          ("f32 persistent 2x4w", {"GANMF_MFMA": "f32", "GANMF_TUNE": "persist=1"}),
                          ("bf16x3 one-tile", {"GANMF_MFMA": "bf16x3", "GANMF_TUNE": "persist=0"})]:
            ms = run(M, N, K, env)
            tf = 2.0 * M * N * K / ms / 1e9
            print("%6d x %6d x %5d  %-18s %8.1f us  %6.1f TF/s  %.3f of 157.3" % (M, N, K, name, ms * 1e3, tf, tf / 157.3), flush=True)
