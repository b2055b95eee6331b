"""BASELINE configs[0] (LastFM 1884 x 17632, 0.22 % dense) on one MI355X: steps/s with the generator step's sparse-aware
real path (CSR row-sum encode, no densify of X) and with the dense path, at the reference's defaults
(k = 10, emb_dim = 32, B = 32: GANMF.py:88-90) and at its tuned LastFM parameters (k = 67, emb_dim = 398, B = 1024).
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
    for sparse in ("1", "0"):
        os.environ["GANMF_SPARSE"] = sparse
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
        g_front = [r for r in rows if r["name"].startswith("densify")]
        enc = [r for r in rows if r["name"].startswith("gemm_encode")]
        print("C1 %-8s k=%d e=%d B=%d  generator real path %-6s: %8.0f steps/s (%.1f us/step); front kernel %.1f us/launch, encode GEMM %.1f us/launch"
              % (name, k, e, B, "sparse" if sparse == "1" else "dense", steps / best, best / steps * 1e6,
                 g_front[0]["ms"] / g_front[0]["launches"] * 1e3, enc[0]["ms"] / enc[0]["launches"] * 1e3), flush=True)
        eng.close()
