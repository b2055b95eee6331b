#!/usr/bin/env python3
"""Per-config evidence (BASELINE.json configs[0], [2], [3]-shard, [4]; configs[1] is bench.py itself): steps/s, the per-class
kernel table of the library's own profiler (kernel start / end events) and a roofline line per class and for the whole step --
algorithmic FLOPs / time against the fp32 MFMA peak, algorithmic bytes / time against HBM -- as markdown on stdout.
usage: python tools/config_profiles.py [name ...]      names: c1_defaults c1_tuned c3 c4_e1024 c4_e32 c5 c5_f16"""
import json
import os
import sys
import time

import numpy as np
import scipy.sparse as sps

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ganmf_amd import _lib as L  # noqa: E402
from ganmf_amd.engine import Engine  # noqa: E402
from ganmf_amd.synthetic import glorot_params, synthetic_urm  # noqa: E402

PEAK_TF, PEAK_GBS = 157.3, 8000.0
HP = dict(d_lr=1e-4, g_lr=1e-4, d_reg=1e-4, g_reg=0.0, m=10.0, recon_coefficient=0.05)


def lastfm():
    return sps.load_npz(os.path.join(ROOT, "tests", "golden", "LastFM_URM_train.npz")).tocsr().astype(np.float32)


CONFIGS = {
    # name: (title, urm factory, k, e, B, engine kwargs, slices)
    "c1_defaults": ("configs[0] GANMF --user, LastFM 1884 x 17632, reference defaults k=10 e=32 B=32", lastfm, 10, 32, 32, {}, 58),
    "c1_tuned": ("configs[0] GANMF --user, LastFM, tuned k=67 e=398 B=1024 (sparse-aware real path)", lastfm, 67, 398, 1024, {}, 1),
    "c3": ("configs[2] GANMF --item, hetrec2011 shape 10109 x 2113, k=100 e=748 B=128", lambda: synthetic_urm(10109, 2113, 0.032, seed=1337), 100, 748, 128, {}, 48),
    "c4_e1024": ("configs[3] one rank's shard 25000 x 50000, k=250 e=1024 B=128", lambda: synthetic_urm(25000, 50000, 0.01, seed=1337), 250, 1024, 128, {}, 24),
    "c4_e32": ("configs[3] one rank's shard 25000 x 50000, k=250 e=32 B=128", lambda: synthetic_urm(25000, 50000, 0.01, seed=1337), 250, 32, 128, {}, 48),
    "c5": ("configs[4] DisGANMF, ML-1M shape 6040 x 3706, k=250 d_nodes=1024 one linear layer B=128, fp32-accurate arithmetic", lambda: synthetic_urm(6040, 3706, 0.035, seed=1337), 250, 1024, 128, dict(model=L.MODEL_DISGANMF, d_layers=1, d_act="linear"), 47),
    "c5_f16": ("configs[4] DisGANMF as written: fp16 MFMA, fp32 Adam, float(uid) column fp32", lambda: synthetic_urm(6040, 3706, 0.035, seed=1337), 250, 1024, 128, dict(model=L.MODEL_DISGANMF, d_layers=1, d_act="linear", mfma="f16"), 47),
}


def run(name):
    title, make, k, e, B, kw, slices = CONFIGS[name]
    urm = make()
    U, N = urm.shape
    eng = Engine(U, N, k, e, B, **dict(HP, **kw))
    eng.set_urm(urm)
    if kw.get("model") == L.MODEL_DISGANMF:
        rng = np.random.RandomState(1337)
        g = lambda a, b: rng.uniform(-np.sqrt(6.0 / (a + b)), np.sqrt(6.0 / (a + b)), size=(a, b)).astype(np.float32)
        for tid, w in ((0, g(N + 1, e)), (1, np.zeros(e, np.float32)), (2, g(e, 1)), (3, np.zeros(1, np.float32)), (100, g(U, k)), (101, g(N, k))):
            eng.set_tensor(tid, w)
    else:
        w = glorot_params(U, N, k, e, seed=1337)
        for n, tid in {"We": 0, "be": 1, "Wd": 2, "bd": 3, "U": 100, "V": 101}.items():
            eng.set_tensor(tid, w[n])
    slices = max(1, min(slices, U // B))
    perm = np.random.RandomState(0).permutation(U)[:B * slices] if slices > 1 else np.random.RandomState(0).permutation(U)
    steps = 2 * -(-len(perm) // B)
    eng.train_epoch(perm, 1, 1)
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        eng.train_epoch(perm, 1, 1)
        best = min(best, time.perf_counter() - t0)
    eng.profile(True)
    eng.train_epoch(perm, 1, 1)
    rows = eng.profile_read()
    eng.profile(False)
    eng.close()
    pairs = steps / 2
    tot_ms = sum(r["ms"] for r in rows)
    tot_fl = sum(r["flops"] for r in rows)
    print("## %s\n" % title)
    print("`tools/config_profiles.py %s`: **%.0f steps/s** (%.1f us per step; D+G pair %.1f us wall, %.1f us of kernels, %d launches per pair)\n"
          % (name, steps / best, best / steps * 1e6, best / pairs * 1e6, tot_ms / pairs * 1e3, round(sum(r["launches"] for r in rows) / pairs)))
    print("| class | launches per pair | us per launch | share | algorithmic TFLOP/s | of fp32 MFMA peak | algorithmic GB/s | of HBM peak |")
    print("|---|---|---|---|---|---|---|---|")
    for r in sorted(rows, key=lambda r: -r["ms"]):
        us = r["ms"] / r["launches"] * 1e3
        tf = r["flops"] / r["ms"] / 1e9 if r["flops"] else 0
        gb = r["bytes"] / r["ms"] / 1e6 if r["bytes"] else 0
        print("| %s | %.1f | %.1f | %.1f %% | %s | %s | %s | %s |" % (
            r["name"], r["launches"] / pairs, us, 100 * r["ms"] / tot_ms, "%.1f" % tf if tf else "-", "%.3f" % (tf / PEAK_TF) if tf else "-",
            "%.0f" % gb if gb else "-", "%.3f" % (gb / PEAK_GBS) if gb else "-"))
    print("\nWhole step: %.2f GFLOP per D+G pair / %.1f us of kernels = **%.1f TFLOP/s = %.3f of the fp32 MFMA peak** (%.1f TFLOP/s on the wall clock).\n"
          % (tot_fl / pairs / 1e9, tot_ms / pairs * 1e3, tot_fl / tot_ms / 1e9, tot_fl / tot_ms / 1e9 / PEAK_TF, tot_fl / pairs / (best / pairs) / 1e12))
    return {"name": name, "steps_per_s": steps / best}


if __name__ == "__main__":
    names = sys.argv[1:] or list(CONFIGS)
    print("# Per-config step profiles (MI355X, one GPU)\n")
    print("Peaks: fp32 MFMA 157.3 TFLOP/s, HBM 8 TB/s (MI355X_MICROARCH.md).  FLOPs and bytes are the ALGORITHMIC ones of each class "
          "(2MNK per GEMM; operands once + results once, the six Adam streams for fused launches), times are the kernels' own start / end "
          "events.\n")
    out = [run(n) for n in names]
    sys.stderr.write(json.dumps(out) + "\n")
