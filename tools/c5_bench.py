"""Steps/s of BASELINE configs[4] -- DisGANMF on ML-1M shape (6040 x 3706, k=250, d_nodes=1024, one linear layer, B=128) -- in the
three arithmetic modes: fp32-accurate (default), fp16 MFMA as the config is written, bf16 MFMA.
Usage: python tools/c5_bench.py [--mode auto,f16,bf16] [--layers L] [--act linear|tanh|relu|sigmoid] [--profile | --profile-all]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ganmf_amd import _lib as L  # noqa: E402
from ganmf_amd.engine import Engine  # noqa: E402
from ganmf_amd.synthetic import synthetic_urm  # noqa: E402
from oracle.ganmf_oracle import DisGANMFOracle  # noqa: E402   (initial weights only: Glorot draws in the reference's tensor order)

U, N, k, e, B = 6040, 3706, 250, 1024, 128
LAYERS = int(sys.argv[sys.argv.index("--layers") + 1]) if "--layers" in sys.argv else 1
ACT = sys.argv[sys.argv.index("--act") + 1] if "--act" in sys.argv else "linear"
hp = dict(d_lr=1e-4, g_lr=5.665e-4, d_reg=3.002e-5, g_reg=0.0, recon_coefficient=0.5)
urm = synthetic_urm(U, N, 0.035, seed=1337)
o = DisGANMFOracle(U, N, k, d_layers=LAYERS, d_nodes=e, d_hidden_act=ACT, dtype=np.float32, seed=1337, **hp)
IDS = {"U": 100, "V": 101, "Wo": 2 * LAYERS, "bo": 2 * LAYERS + 1}
for _l in range(LAYERS):
    IDS["W%d" % _l], IDS["b%d" % _l] = 2 * _l, 2 * _l + 1
perm = np.random.RandomState(0).permutation(U)[:B * 47]
MODES = [None if m == "auto" else m for m in sys.argv[sys.argv.index("--mode") + 1].split(",")] if "--mode" in sys.argv else [None, "f16", "bf16"]
for mfma in MODES:
    eng = Engine(U, N, k, e, B, model=L.MODEL_DISGANMF, d_layers=LAYERS, d_act=ACT, m=0.0, mfma=mfma, **hp)
    eng.set_urm(urm)
    for n, tid in IDS.items():
        eng.set_tensor(tid, o.p[n])
    eng.train_epoch(perm[:B * 8], 1, 1)
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        eng.train_epoch(perm, 1, 1)
        best = min(best, time.perf_counter() - t0)
    print("C5 DisGANMF mfma=%-5s: %7.0f steps/s (%.1f us/step)" % (mfma or "auto", 94 / best, best / 94 * 1e6))
    eng.close()
    if (mfma is None and "--profile" in sys.argv) or "--profile-all" in sys.argv:
        eng2 = Engine(U, N, k, e, B, model=L.MODEL_DISGANMF, d_layers=LAYERS, d_act=ACT, m=0.0, mfma=mfma, **hp)
        eng2.set_urm(urm)
        for n, tid in IDS.items():
            eng2.set_tensor(tid, o.p[n])
        eng2.train_epoch(perm[:B * 8], 1, 1)
        eng2.profile(True)
        eng2.train_epoch(perm[:B * 24], 1, 1)
        rows = eng2.profile_read()
        tot = sum(r["ms"] for r in rows)
        for r in sorted(rows, key=lambda r: -r["ms"]):
            print("   %-52s %4d launches %7.1f us/launch %5.1f %%" % (r["name"][:52], r["launches"], r["ms"] / r["launches"] * 1e3, 100 * r["ms"] / tot))
        eng2.close()
