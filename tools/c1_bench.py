"""BASELINE configs[0] (LastFM 1884 x 17632, 0.22 % dense) on one MI355X: steps/s with the sparse-aware real path (SURVEY
8f-3) of the generator step (CSR row-sum encode, no densify of X), of the discriminator step as well (CSR lookup in the decode
epilogue, X^T.dE_r from the CSC matrix in the encoder-gradient epilogue) and with the dense path, at the reference's defaults
(k = 10, emb_dim = 32, B = 32: GANMF.py:88-90) and at its tuned LastFM parameters (k = 67, emb_dim = 398, B = 1024); the
last line of each block is the planner's own choice.
Usage: python tools/c1_bench.py"""
import json
import os
import sys
import time

import numpy as np
import scipy.sparse as sps

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ganmf_amd.engine import Engine  # noqa: E402
from ganmf_amd.synthetic import glorot_params  # noqa: E402

urm = sps.load_npz(os.path.join(ROOT, "tests", "golden", "LastFM_URM_train.npz")).tocsr().astype(np.float32)
U, N = urm.shape
tuned = json.load(open(os.path.join(ROOT, "tests", "golden", "statistical_kat_lastfm_user.json")))["best_params"]
print("LastFM train split %d x %d, %d stored entries (%.3f %% dense)" % (U, N, urm.nnz, 100.0 * urm.nnz / (U * N)))
for name, k, e, B in (("defaults", 10, 32, 32), ("tuned", tuned["num_factors"], tuned["emb_dim"], tuned["batch_size"])):
    for sparse, sparse_d, label in (("0", "0", "dense G, dense D"), ("1", "0", "sparse G, dense D"), ("1", "1", "sparse G, sparse D"),
                                    (None, None, "planner")):
        for var, val in (("GANMF_SPARSE", sparse), ("GANMF_SPARSE_D", sparse_d)):
            if val is None:
                os.environ.pop(var, None)
            else:
                os.environ[var] = val
        hp = dict(d_lr=1e-4, g_lr=1e-4, d_reg=1e-5, g_reg=0.0, m=10.0, recon_coefficient=0.3)
        eng = Engine(U, N, k, e, B, **hp)
        eng.set_urm(urm)
        w = glorot_params(U, N, k, e, seed=1337)
        for n, tid in {"We": 0, "be": 1, "Wd": 2, "bd": 3, "U": 100, "V": 101}.items():
            eng.set_tensor(tid, w[n])
        perm = np.random.RandomState(0).permutation(U)
        eng.train_epoch(perm, 1, 1)
        steps = 2 * -(-U // B)
        best = 1e9
        for _ in range(5):
            t0 = time.perf_counter()
            eng.train_epoch(perm, 1, 1)
            best = min(best, time.perf_counter() - t0)
        eng.profile(True)
        eng.train_epoch(perm, 1, 1)
        rows = eng.profile_read()
        eng.profile(False)
        def per(prefix):
            r = [x for x in rows if x["name"].startswith(prefix)]
            return sum(x["ms"] for x in r) / max(sum(x["launches"] for x in r), 1) * 1e3 if r else float("nan")
        print("C1 %-8s k=%d e=%d B=%d  %-18s: %8.0f steps/s (%.1f us/step); us/launch: encode %.1f  decode %.1f  gWd+gWe(+Adam) %.1f / gWe %.1f"
              % (name, k, e, B, label, steps / best, best / steps * 1e6, per("gemm_encode"), per("gemm_decode"),
                 per("gemm_gWd + gemm_gWe"), per("gemm_gWe")), flush=True)
        eng.close()
