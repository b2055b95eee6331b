#!/usr/bin/env python3
"""Fixed cost of one ganmf_train_epoch call at C2: wall time of calls with 2 .. 47 slices (1 D + 1 G pass each), least-squares
T = a + b * steps.  usage: python tools/epoch_overhead.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ganmf_amd.engine import Engine  # noqa: E402
from ganmf_amd.synthetic import glorot_params, synthetic_urm  # noqa: E402

U, N, k, e, B = 6040, 3706, 250, 992, 128
hp = dict(d_lr=1e-4, g_lr=1e-4, d_reg=1e-4, g_reg=0.0, m=10.0, recon_coefficient=0.05)
urm = synthetic_urm(U, N, 0.045, seed=1337)
w = glorot_params(U, N, k, e, seed=1337)
eng = Engine(U, N, k, e, B, **hp)
eng.set_urm(urm)
for n, tid in {"We": 0, "be": 1, "Wd": 2, "bd": 3, "U": 100, "V": 101}.items():
    eng.set_tensor(tid, w[n])
perm = np.random.RandomState(0).permutation(U).astype(np.int32)
eng.train_epoch(perm[:B * 47], 1, 1)
xs, ys = [], []
for slices in (2, 5, 10, 20, 47, 2, 5, 10, 20, 47):
    p = perm[:B * slices]
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        eng.train_epoch(p, 1, 1)
        best = min(best, time.perf_counter() - t0)
    xs.append(2 * slices); ys.append(best * 1e6)
    print("%2d slices: %8.1f us  (%.1f us/step)" % (slices, best * 1e6, best * 1e6 / (2 * slices)), flush=True)
b, a = np.polyfit(xs, ys, 1)
print("T = %.1f us + %.2f us/step" % (a, b))
eng.close()
