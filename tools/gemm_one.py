"""One GEMM through ganmf_gemm_f32 for profiling: python tools/gemm_one.py LAYOUT M N K [tile] [nsplit] [iters]
(kernel variant / arithmetic from the environment: GANMF_MFMA, GANMF_TUNE=persist=..,ring=..)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ganmf_amd.engine import gemm_f32  # noqa: E402

if __name__ == "__main__":
    layout, M, N, K = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    tile = int(sys.argv[5]) if len(sys.argv) > 5 else 0
    nsplit = int(sys.argv[6]) if len(sys.argv) > 6 else 0
    iters = int(sys.argv[7]) if len(sys.argv) > 7 else 5
    akm, bkm = {"NT": (False, False), "NN": (False, True), "TN": (True, True)}[layout]
    rng = np.random.RandomState(0)
    A = rng.standard_normal((K, M) if akm else (M, K)).astype(np.float32)
    B = rng.standard_normal((K, N) if bkm else (N, K)).astype(np.float32)
    _, ms = gemm_f32(A, B, akm, bkm, tile=tile, nsplit=nsplit, iters=iters)
    print("%s %dx%dx%d: %.2f us, %.1f TF/s" % (layout, M, N, K, ms * 1e3, 2.0 * M * N * K / ms / 1e9))
