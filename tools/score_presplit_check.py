"""The scoring product on the pre-split persistent kernel (gemm_bf16p.hpp) against float64 and against the one-tile split-bf16
kernel, at shapes around its edges (K of one K-tile, K beyond eight K-tiles, ragged M / N, both orientations), with timings.
usage: python tools/score_presplit_check.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ganmf_amd.engine import Engine  # noqa: E402

rng = np.random.RandomState(0)
for (U, N, k) in [(6040, 3706, 250), (4001, 4100, 7), (3000, 9000, 64), (5000, 5000, 300), (6040, 3706, 33)]:
    Uf = (rng.standard_normal((U, k)) * np.exp(rng.uniform(-3, 3, (U, 1)))).astype(np.float32)
    Vf = (rng.standard_normal((N, k)) * np.exp(rng.uniform(-3, 3, (N, 1)))).astype(np.float32)
    ref = Uf.astype(np.float64) @ Vf.astype(np.float64).T
    bound = np.abs(Uf).astype(np.float64) @ np.abs(Vf).astype(np.float64).T      # |a|.|b| per element
    out = {}
    for pre in ("1", "0"):
        os.environ["GANMF_SCORE_PRESPLIT"] = pre
        eng = Engine(U, N, k, 8, 8)
        eng.set_tensor(100, Uf)
        eng.set_tensor(101, Vf)
        ids = rng.permutation(U)[: U - 3]
        s = eng.scores(ids)
        st = eng.scores(np.arange(N - 5), transposed=True)
        err = np.max(np.abs(s - ref[ids]) / bound[ids])
        errt = np.max(np.abs(st - ref.T[: N - 5]) / bound.T[: N - 5])
        ms = eng.bench_scores(U, iters=50)
        ms = eng.bench_scores(U, iters=50)
        out[pre] = (s, st)
        print("%5d x %5d x %3d  presplit=%s  max err / (|a|.|b|) %.2e (item mode %.2e)   %7.1f us  %6.1f TFLOP/s" % (
            U, N, k, pre, err, errt, ms * 1e3, 2.0 * U * N * k / ms / 1e9), flush=True)
        assert err < 4e-7 and errt < 4e-7, (err, errt)
        eng.close()
    d = np.max(np.abs(out["1"][0] - out["0"][0]) / bound[ids])
    print("      presplit vs one-tile kernel: max difference / (|a|.|b|) %.2e" % d)
