"""Steps/s of one rank's shard of BASELINE configs[3] (200 k x 50 k over 8 GPUs -> 25 k x 50 k per GPU, k=250,
e=1024 and e=32, B=128) on a single MI355X, with the per-kernel table.  Usage: python tools/c4_bench.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ganmf_amd.engine import Engine  # noqa: E402
from ganmf_amd.synthetic import glorot_params, synthetic_urm  # noqa: E402

U, N, k, B = 25000, 50000, 250, 128
urm = synthetic_urm(U, N, 0.01, seed=1337)
for e in (1024, 32):
    hp = dict(d_lr=1e-4, g_lr=1e-4, d_reg=1e-4, g_reg=0.0, m=10.0, recon_coefficient=0.01)
    eng = Engine(U, N, k, e, B, **hp)
    eng.set_urm(urm)
    w = glorot_params(U, N, k, e, seed=1337)
    for n, tid in {"We": 0, "be": 1, "Wd": 2, "bd": 3, "U": 100, "V": 101}.items():
        eng.set_tensor(tid, w[n])
    perm = np.random.RandomState(0).permutation(U)[:B * 96]
    eng.train_epoch(perm[:B * 16], 1, 1)
    t0 = time.perf_counter()
    eng.train_epoch(perm, 1, 1)
    dt = time.perf_counter() - t0
    print("C4 shard e=%d: %.1f steps/s (%.1f us/step, D+G pair %.1f us)" % (e, 192 / dt, dt / 192 * 1e6, dt / 96 * 1e6))
    eng.profile(True)
    eng.train_epoch(perm[:B * 32], 1, 1)
    rows = eng.profile_read()
    eng.profile(False)
    tot = sum(r["ms"] for r in rows)
    for r in sorted(rows, key=lambda r: -r["ms"])[:12]:
        print("   %-28s %5d launches %8.1f us/launch %5.1f%%  %6.1f TFLOP/s %7.1f GB/s" % (
            r["name"], r["launches"], r["ms"] / r["launches"] * 1e3, 100 * r["ms"] / tot,
            r["flops"] / (r["ms"] * 1e-3) / 1e12 if r["ms"] else 0, r["bytes"] / (r["ms"] * 1e-3) / 1e9 if r["ms"] else 0))
    eng.close()
