"""Arithmetic modes of the GEMM K loops (include/ganmf_hip.h GANMF_FLAG_MFMA_*, csrc/gemm_f32.hpp MfmaMode).

  bf16x3  every fp32 operand split exactly into three bf16 pieces, six piece products accumulated in fp32 on the
          bf16 matrix cores: must be as accurate as the fp32 MFMA path (same bound as tests/test_gpu_gemm.py).
  bf16    operands rounded to one bf16, fp32 accumulate, fp32 master weights / Adam — the mixed-precision variant
          BASELINE configs[4] asks for: tolerances are those of an 8-bit mantissa and are stated per assertion.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

from ganmf_amd.synthetic import glorot_params, synthetic_urm
from oracle.ganmf_oracle import DisGANMFOracle, GANMFOracle

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _err(got, ref):
    return np.max(np.abs(np.asarray(got, np.float64).reshape(np.shape(ref)) - ref)) / (np.max(np.abs(ref)) + 1e-30)


_GEMM_SNIPPET = r"""
import numpy as np, sys, json
sys.path.insert(0, %r)
from ganmf_amd.engine import gemm_f32
rng = np.random.RandomState(4)
out = {}
for name, (akm, bkm, M, N, K, tile) in {"NT": (0, 0, 300, 517, 250, 128), "NN": (0, 1, 256, 992, 3707, 64),
                                         "TN": (1, 1, 993, 640, 256, 128), "NTbig": (0, 0, 1500, 1100, 250, 0)}.items():
    A = (rng.standard_normal((K, M) if akm else (M, K)) * np.exp(rng.uniform(-12, 4, size=(K, M) if akm else (M, K)))).astype(np.float32)
    B = rng.standard_normal((K, N) if bkm else (N, K)).astype(np.float32)
    C, _ = gemm_f32(A, B, bool(akm), bool(bkm), tile=tile)
    a = (A.T if akm else A).astype(np.float64); b = (B if bkm else B.T).astype(np.float64)
    ref = a @ b
    bound = np.abs(a) @ np.abs(b)
    out[name] = float(np.max(np.abs(C - ref) / (bound + 1e-300)))
    # exact cases: identity operand reproduces the other operand bit for bit
    if name == "NT":
        I = np.eye(64, dtype=np.float32)
        Bm = (rng.standard_normal((200, 64)) * np.exp(rng.uniform(-30, 30, size=(200, 64)))).astype(np.float32)
        Ci, _ = gemm_f32(I, Bm, False, False)
        out["identity_exact"] = bool(np.array_equal(Ci, Bm.T))
print(json.dumps(out))
"""


@pytest.mark.parametrize("mode,bound", [("f32", 1e-6), ("bf16x3", 1e-6), ("bf16", 8e-3)])
def test_gemm_error_bounds_per_mode(mode, bound):
    """max |C - A.B| / (|A|.|B|): the two fp32-accurate paths share ONE bound (a few fp32 roundings of the exact
    product sum, operands spread over 16 decades); the single-bf16 path within 2^-7.  GANMF_MFMA selects the mode of ganmf_gemm_f32, so
    each mode runs in its own process."""
    import json
    env = dict(os.environ, GANMF_MFMA=mode)
    r = subprocess.run([sys.executable, "-c", _GEMM_SNIPPET % ROOT], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    res = json.loads(r.stdout.strip().splitlines()[-1])
    for k in ("NT", "NN", "TN", "NTbig"):
        assert res[k] <= bound, (mode, k, res[k])
    if mode != "bf16":
        assert res["identity_exact"] is True
    else:
        assert res["NT"] > 1e-5      # the mode really rounds to bf16


def test_ganmf_steps_all_modes_vs_oracle():
    """The same two D+G updates at ML-1M shape under "auto" (split-bf16 / fp32 per GEMM), forced fp32 and the
    mixed-precision bf16 mode."""
    from ganmf_amd.engine import Engine
    U, N, k, e, B = 6040, 3706, 250, 992, 128
    hp = dict(d_lr=1e-4, g_lr=1.6532e-4, d_reg=1e-4, g_reg=0.0, m=10.0, recon_coefficient=0.01)
    urm = synthetic_urm(U, N, 0.035, seed=11)
    w = glorot_params(U, N, k, e, seed=5)
    ids = {"We": 0, "be": 1, "Wd": 2, "bd": 3, "U": 100, "V": 101}
    perm = np.random.RandomState(3).permutation(U)
    o = GANMFOracle(U, N, k, e, dtype=np.float64, **hp)
    o.set_params(**w)
    ref_losses = []
    for t in range(2):
        uids = perm[t * B:(t + 1) * B]
        X = urm[uids].toarray()
        ref_losses.append((o.d_step(uids, X), o.g_step(uids, X)))
    ref_scores = o.scores(perm[:2048])
    # parameter bound 1e-4 for the fp32-accurate modes: after two updates the decoder bias (|g| ~ Adam's epsilon, so the
    # update divides by |g| + eps) is 3e-5 off the fp64 oracle in the numpy-fp32 restatement itself, 4.6e-5 with the
    # fp32 MFMA everywhere and 5.2e-5 with the split-bf16 loop on the two weight-gradient GEMMs; every other tensor
    # agrees to 1e-6 .. 4e-6 in all three
    for mfma, ltol, ptol, stol in ((None, 5e-5, 1e-4, 1e-4), ("f32", 5e-5, 1e-4, 1e-4), ("bf16", 5e-3, 2e-2, 1e-2)):   # bf16: an Adam step is +-lr whatever |g|, so a near-zero gradient
                                                # whose sign flips under rounding moves theta by 2 lr T = 1 % of max|We|
        eng = Engine(U, N, k, e, B, mfma=mfma, **hp)
        eng.set_urm(urm)
        for n, tid in ids.items():
            eng.set_tensor(tid, w[n])
        for t in range(2):
            uids = perm[t * B:(t + 1) * B]
            ld, lg = eng.train_step(0, uids), eng.train_step(1, uids)
            assert abs(ld - ref_losses[t][0]) <= ltol * abs(ref_losses[t][0]), (mfma, t, ld, ref_losses[t][0])
            assert abs(lg - ref_losses[t][1]) <= ltol * abs(ref_losses[t][1]), (mfma, t, lg, ref_losses[t][1])
        for n, tid in ids.items():
            tol = ptol
            if mfma == "bf16":      # |delta theta| <= 2 lr per update (sign flip of a near-zero gradient)
                tol = 2.2 * (hp["g_lr"] if n in ("U", "V") else hp["d_lr"]) * 2 / np.max(np.abs(o.p[n])) + 1e-3
            assert _err(eng.get_tensor(tid), o.p[n]) <= tol, (mfma, n)
        # 2048 score rows: "auto" runs this GEMM on the split-bf16 kernel
        assert _err(eng.scores(perm[:2048]), ref_scores) <= stol, mfma
        eng.close()


def test_c5_disganmf_mixed_precision():
    """BASELINE configs[4]: DisGANMF at ML-1M shape with low-precision MFMA inputs and fp32 Adam accumulators.
    bf16 stands in for fp16 (same matrix-core rate on gfx950; fp16's 5-bit exponent would need loss scaling for
    the ~1e-6 gradients).  The raw float(uid) feature (DisGANMF.py:110) is rounded to 8 bits as well, so logits of
    O(100) move by O(0.5): losses agree to 3 %, first-moment (gradient) tensors to 5 % of their scale."""
    from ganmf_amd import _lib as L
    from ganmf_amd.engine import Engine
    U, N, k, e, B = 6040, 3706, 250, 1024, 128
    hp = dict(d_lr=1e-4, g_lr=5.665e-4, d_reg=3.002e-5, g_reg=0.0, recon_coefficient=0.5)
    urm = synthetic_urm(U, N, 0.035, seed=12)
    o = DisGANMFOracle(U, N, k, d_layers=1, d_nodes=e, d_hidden_act="linear", dtype=np.float64, seed=7, **hp)
    eng = Engine(U, N, k, e, B, model=L.MODEL_DISGANMF, d_layers=1, d_act="linear", m=0.0, mfma="bf16", **hp)
    eng.set_urm(urm)
    ids = {"W0": 0, "b0": 1, "Wo": 2, "bo": 3, "U": 100, "V": 101}
    for n, tid in ids.items():
        eng.set_tensor(tid, o.p[n])
    perm = np.random.RandomState(1).permutation(U)
    uids = perm[:B]
    X = urm[uids].toarray()
    ld_ref, ld = o.d_step(uids, X), eng.train_step(0, uids)
    lg_ref, lg = o.g_step(uids, X), eng.train_step(1, uids)
    assert abs(ld - ld_ref) <= 3e-2 * abs(ld_ref), (ld, ld_ref)
    assert abs(lg - lg_ref) <= 3e-2 * abs(lg_ref), (lg, lg_ref)
    for n, tid in ids.items():
        m_ref = (o.opt_d.slots[n] if n in o.opt_d.slots else o.opt_g.slots[n])[0]
        assert _err(eng.get_tensor(tid, slot=L.SLOT_ADAM_M), m_ref) <= 5e-2, n
        lr = hp["g_lr"] if n in ("U", "V") else hp["d_lr"]
        assert _err(eng.get_tensor(tid), o.p[n]) <= 2.2 * lr / np.max(np.abs(o.p[n])) + 1e-3, n   # <= 2 lr per update
    # master weights and moments are float32: a second identical engine reproduces them bit for bit
    eng2 = Engine(U, N, k, e, B, model=L.MODEL_DISGANMF, d_layers=1, d_act="linear", m=0.0, mfma="bf16", **hp)
    eng2.set_urm(urm)
    o2 = DisGANMFOracle(U, N, k, d_layers=1, d_nodes=e, d_hidden_act="linear", dtype=np.float64, seed=7, **hp)
    for n, tid in ids.items():
        eng2.set_tensor(tid, o2.p[n])
    eng2.train_step(0, uids); eng2.train_step(1, uids)
    for n, tid in ids.items():
        assert np.array_equal(eng.get_tensor(tid), eng2.get_tensor(tid)), n
    eng.close(); eng2.close()
