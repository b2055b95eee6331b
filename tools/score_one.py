"""bench_scores at ML-1M shape (for rocprofv3 traces): python tools/score_one.py [iters]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ganmf_amd.engine import Engine  # noqa: E402

U, N, k = 6040, 3706, 250
rng = np.random.RandomState(0)
eng = Engine(U, N, k, 8, 8)
eng.set_tensor(100, rng.standard_normal((U, k)).astype(np.float32))
eng.set_tensor(101, rng.standard_normal((N, k)).astype(np.float32))
it = int(sys.argv[1]) if len(sys.argv) > 1 else 100
eng.bench_scores(U, iters=it)
ms = eng.bench_scores(U, iters=it)
print("scores %d x %d x %d: %.1f us per product (%.1f TFLOP/s)" % (U, N, k, ms * 1e3, 2.0 * U * N * k / ms / 1e9))
eng.close()
