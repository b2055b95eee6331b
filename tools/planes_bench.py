"""The generator step's four GEMMs at the ML-1M shape on the 16-wave split-bf16 plan: operands split inside the K loop (gemm_bf16k.hpp)
against operands split ahead of the launch (gemm_planes.hpp); launch + split-K reduce, averaged over `iters` back-to-back runs."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ganmf_amd.engine import gemm_f32  # noqa: E402

os.environ["GANMF_MFMA"] = "f32"
os.environ["GANMF_X3KG"] = "3"
rng = np.random.RandomState(0)
for name, (M, N, K), bkm, ns in (("encode D [256,3707]x[3707,992]", (256, 992, 3707), True, 4), ("encode/dec G [128,993]x[993,3706]", (128, 3706, 993), True, 2),
                                 ("dE G [128,3706]x[992,3706]^T", (128, 992, 3706), False, 8), ("dF [128,992]x[3706,992]^T", (128, 3706, 992), False, 2),
                                 ("decode D [256,993]x[993,3706]", (256, 3706, 993), True, 1), ("dE D [256,3706]x[992,3706]^T", (256, 992, 3706), False, 4)):
    A = rng.standard_normal((M, K)).astype(np.float32)
    B = rng.standard_normal((K, N) if bkm else (N, K)).astype(np.float32)
    out = []
    for planes in (0, 1):
        os.environ["GANMF_TUNE"] = "kg=4,ring=3,planes=%d" % planes
        gemm_f32(A, B, False, bkm, tile=64, nsplit=ns, iters=50)
        _, ms = gemm_f32(A, B, False, bkm, tile=64, nsplit=ns, iters=200)
        out.append(ms * 1e3)
    print("%-36s split in the loop %6.2f us   pre-split planes %6.2f us   (%+.1f %%)" % (name, out[0], out[1], 100 * (out[1] / out[0] - 1)))
