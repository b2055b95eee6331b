"""Steps/s and the per-class kernel table of a GANMF step at an arbitrary shape (synthetic URM of the given density).
usage: python tools/config_bench.py U N k e B [density]      e.g. C3 (hetrec item): 10109 2113 100 748 128 0.02"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ganmf_amd.engine import Engine  # noqa: E402
from ganmf_amd.synthetic import glorot_params, synthetic_urm  # noqa: E402

U, N, k, e, B = [int(x) for x in sys.argv[1:6]]
dens = float(sys.argv[6]) if len(sys.argv) > 6 else 0.03
hp = dict(d_lr=1e-4, g_lr=1e-4, d_reg=1e-4, g_reg=0.0, m=10.0, recon_coefficient=0.05)
urm = synthetic_urm(U, N, dens, seed=1337)
w = glorot_params(U, N, k, e, seed=1337)
eng = Engine(U, N, k, e, B, **hp)
eng.set_urm(urm)
for n, tid in {"We": 0, "be": 1, "Wd": 2, "bd": 3, "U": 100, "V": 101}.items():
    eng.set_tensor(tid, w[n])
slices = min(U // B, 48)
perm = np.random.RandomState(0).permutation(U)[:B * slices]
eng.train_epoch(perm[:B * min(slices, 8)], 1, 1)
best = 1e9
for _ in range(3):
    t0 = time.perf_counter()
    eng.train_epoch(perm, 1, 1)
    best = min(best, time.perf_counter() - t0)
print("U=%d N=%d k=%d e=%d B=%d: %.0f steps/s (%.1f us/step, D+G pair %.1f us)" % (U, N, k, e, B, 2 * slices / best, best / (2 * slices) * 1e6, best / slices * 1e6))
eng.profile(True)
eng.train_epoch(perm[:B * min(slices, 24)], 1, 1)
rows = eng.profile_read()
tot = sum(r["ms"] for r in rows)
for r in sorted(rows, key=lambda r: -r["ms"]):
    print("   %-52s %4d launches %7.1f us/launch %5.1f %%" % (r["name"][:52], r["launches"], r["ms"] / r["launches"] * 1e3, 100 * r["ms"] / tot))
eng.close()
