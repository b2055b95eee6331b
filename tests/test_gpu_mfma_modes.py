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


@pytest.mark.parametrize("tune", ["", "dis_top_gw=0", "dis_top_gw=0,dis_uid_top=0"])
@pytest.mark.parametrize("mfma,loss_tol,m_tol", [("f16", 2e-5, 2e-3), ("bf16", 2e-5, 1.2e-2)])
def test_c5_disganmf_mixed_precision(mfma, loss_tol, m_tol, tune, monkeypatch):
    """BASELINE configs[4]: DisGANMF at ML-1M shape, k = 250, fp16 MFMA inputs (v_mfma_f32_32x32x16_f16) with fp32
    accumulation, fp32 master weights and fp32 Adam accumulators.  fp16 carries 11 significant bits; operands that carry
    the 1/B of the loss gradient are scaled by a power of two at conversion (static loss scaling per GEMM).  The raw
    float(uid) feature (DisGANMF.py:59,110-111; values up to 6039) is kept in fp32 outside the low-precision K loop, as
    SURVEY 7 prescribes: rank-1 epilogue term forward, fp32 reduction for its weight-row gradient.  Losses agree with
    the fp64 oracle to 2e-5 (measured 3e-8 / 7e-7: the logits are dominated by the fp32 uid term), first-moment
    (= gradient) tensors to 2e-3 of their scale (measured <= 6.4e-4).  The bf16 variant (8 significant bits, same uid
    handling) is held to 2e-5 / 1.2e-2 (measured 8e-7 / 5.5e-3).  Round 1 rounded the uid column too: 3e-2 / 5e-2.
    tune: "" = the round-6 step (dz_0 rows written by the slab-sum launch, output-layer and float(uid)-row column sums + their Adam updates
    inside the gradient GEMM's launch, gemm_bf16s_colsum); dis_top_gw=0 = dis_dz_top_kernel as its own launch (with the uid row in it);
    + dis_uid_top=0 = dis_uid_grad_kernel as well (round 5)."""
    if tune:
        monkeypatch.setenv("GANMF_TUNE", tune)
    from ganmf_amd import _lib as L
    from ganmf_amd.engine import Engine
    U, N, k, e, B = 6040, 3706, 250, 1024, 128
    hp = dict(d_lr=1e-4, g_lr=5.665e-4, d_reg=3.002e-5, g_reg=0.0, recon_coefficient=0.5)
    urm = synthetic_urm(U, N, 0.035, seed=12)
    o = DisGANMFOracle(U, N, k, d_layers=1, d_nodes=e, d_hidden_act="linear", dtype=np.float64, seed=7, **hp)
    eng = Engine(U, N, k, e, B, model=L.MODEL_DISGANMF, d_layers=1, d_act="linear", m=0.0, mfma=mfma, **hp)
    eng.set_urm(urm)
    ids = {"W0": 0, "b0": 1, "Wo": 2, "bo": 3, "U": 100, "V": 101}
    for n, tid in ids.items():
        eng.set_tensor(tid, o.p[n])
    perm = np.random.RandomState(1).permutation(U)
    uids = perm[:B]
    X = urm[uids].toarray()
    ld_ref, ld = o.d_step(uids, X), eng.train_step(0, uids)
    lg_ref, lg = o.g_step(uids, X), eng.train_step(1, uids)
    print("%s: dloss %.6f (oracle %.6f, rel %.1e)  gloss %.6f (oracle %.6f, rel %.1e)" % (
        mfma, ld, ld_ref, abs(ld - ld_ref) / abs(ld_ref), lg, lg_ref, abs(lg - lg_ref) / abs(lg_ref)))
    assert abs(ld - ld_ref) <= loss_tol * abs(ld_ref), (ld, ld_ref)
    assert abs(lg - lg_ref) <= loss_tol * abs(lg_ref), (lg, lg_ref)
    for n, tid in ids.items():
        m_ref = (o.opt_d.slots[n] if n in o.opt_d.slots else o.opt_g.slots[n])[0]
        err = _err(eng.get_tensor(tid, slot=L.SLOT_ADAM_M), m_ref)
        print("   first moment %-3s error %.2e of its scale" % (n, err))
        assert err <= m_tol, (n, err)
        lr = hp["g_lr"] if n in ("U", "V") else hp["d_lr"]
        assert _err(eng.get_tensor(tid), o.p[n]) <= 2.2 * lr / np.max(np.abs(o.p[n])) + 1e-3, n   # <= 2 lr per update
    # master weights and moments are float32: a second identical engine reproduces them bit for bit
    eng2 = Engine(U, N, k, e, B, model=L.MODEL_DISGANMF, d_layers=1, d_act="linear", m=0.0, mfma=mfma, **hp)
    eng2.set_urm(urm)
    o2 = DisGANMFOracle(U, N, k, d_layers=1, d_nodes=e, d_hidden_act="linear", dtype=np.float64, seed=7, **hp)
    for n, tid in ids.items():
        eng2.set_tensor(tid, o2.p[n])
    eng2.train_step(0, uids); eng2.train_step(1, uids)
    for n, tid in ids.items():
        assert np.array_equal(eng.get_tensor(tid), eng2.get_tensor(tid)), n
    eng.close(); eng2.close()


@pytest.mark.parametrize("act,layers", [("tanh", 2), ("relu", 3), ("sigmoid", 1)])
def test_disganmf_f16_hidden_layers(act, layers):
    """fp16 mode through non-linear hidden layers and depth > 1 (the scaled backward GEMMs dz_l . W_l^T and the
    activation-gradient epilogue), small enough for a 3-step trajectory against the fp64 oracle."""
    from ganmf_amd import _lib as L
    from ganmf_amd.engine import Engine
    U, N, k, e, B = 300, 500, 16, 96, 64
    hp = dict(d_lr=1e-3, g_lr=1e-3, d_reg=1e-4, g_reg=0.0, recon_coefficient=0.3)
    urm = synthetic_urm(U, N, 0.05, seed=3)
    o = DisGANMFOracle(U, N, k, d_layers=layers, d_nodes=e, d_hidden_act=act, dtype=np.float64, seed=4, **hp)
    eng = Engine(U, N, k, e, B, model=L.MODEL_DISGANMF, d_layers=layers, d_act=act, m=0.0, mfma="f16", **hp)
    eng.set_urm(urm)
    ids = {}
    for l in range(layers):
        ids["W%d" % l], ids["b%d" % l] = 2 * l, 2 * l + 1
    ids.update({"Wo": 2 * layers, "bo": 2 * layers + 1, "U": 100, "V": 101})
    for n, tid in ids.items():
        eng.set_tensor(tid, o.p[n])
    perm = np.random.RandomState(2).permutation(U)
    for t in range(3):
        uids = perm[t * B:(t + 1) * B]
        X = urm[uids].toarray()
        ld_ref, ld = o.d_step(uids, X), eng.train_step(0, uids)
        lg_ref, lg = o.g_step(uids, X), eng.train_step(1, uids)
        assert abs(ld - ld_ref) <= 5e-3 * abs(ld_ref) + 1e-5, (act, t, ld, ld_ref)
        assert abs(lg - lg_ref) <= 5e-3 * abs(lg_ref) + 1e-5, (act, t, lg, lg_ref)
    for n, tid in ids.items():
        m_ref = (o.opt_d.slots[n] if n in o.opt_d.slots else o.opt_g.slots[n])[0]
        got = np.asarray(eng.get_tensor(tid, slot=L.SLOT_ADAM_M), np.float64).reshape(np.shape(m_ref))
        if act != "relu":
            assert _err(got, m_ref) <= 2e-2, n
            continue
        # relu: a pre-activation within fp16 rounding distance of zero flips its unit on or off -- a discrete change of
        # one sample's gradient that three Adam steps carry into the moments.  Which units sit that close to zero depends
        # on the last bit of the summation order, so a max-norm bound is not a property of the arithmetic: the bound is
        # on how MANY rows are touched (a flip stays inside the rows of its sample / unit) and on the bulk of the tensor.
        scale = np.max(np.abs(m_ref)) + 1e-30
        d = np.abs(got - m_ref).reshape(got.shape[0], -1) / scale if got.ndim > 1 else np.abs(got - m_ref)[None, :] / scale
        row_bad = np.mean(d.max(axis=1) > 2e-2) if got.ndim > 1 else np.mean(d > 2e-2)
        fro = np.linalg.norm(got - m_ref) / (np.linalg.norm(m_ref) + 1e-30)
        print("   relu first moment %-3s: max %.3f of scale, rows/entries beyond 2e-2: %.1f %%, relative Frobenius %.3f" % (
            n, d.max(), 100 * row_bad, fro))
        # measured: relative Frobenius error 0.025 .. 0.042 on every discriminator tensor; a flipped unit shows up as one
        # sample's row of U (max-norm 0.28 of the scale in round 2) or one unit's column of a weight matrix
        assert fro <= 0.12 and d.max() <= 0.6, (n, fro, d.max())
        if n == "U":
            assert row_bad <= 0.10, (n, row_bad)
    eng.close()


def test_ganmf_f16_step_vs_oracle():
    """GANMF (autoencoder discriminator) in fp16 mode: the operands that carry the 2/(B.N) of the MSE gradients (Es, dE,
    dF) are scaled by 2^round(log2(B.N)) at conversion; one D + one G update against the fp64 oracle."""
    from ganmf_amd.engine import Engine
    U, N, k, e, B = 2000, 3706, 250, 992, 128
    hp = dict(d_lr=1e-4, g_lr=1e-3, d_reg=1e-5, g_reg=0.0, m=10.0, recon_coefficient=0.2)
    urm = synthetic_urm(U, N, 0.035, seed=5)
    from oracle.ganmf_oracle import GANMFOracle
    o = GANMFOracle(U, N, k, e, dtype=np.float64, seed=6, **hp)
    eng = Engine(U, N, k, e, B, mfma="f16", **hp)
    eng.set_urm(urm)
    ids = {"We": 0, "be": 1, "Wd": 2, "bd": 3, "U": 100, "V": 101}
    for n, tid in ids.items():
        eng.set_tensor(tid, o.p[n])
    uids = np.random.RandomState(3).permutation(U)[:B]
    X = urm[uids].toarray()
    ld_ref, ld = o.d_step(uids, X), eng.train_step(0, uids)
    lg_ref, lg = o.g_step(uids, X), eng.train_step(1, uids)
    assert abs(ld - ld_ref) <= 2e-3 * abs(ld_ref), (ld, ld_ref)
    assert abs(lg - lg_ref) <= 2e-3 * abs(lg_ref), (lg, lg_ref)
    from ganmf_amd import _lib as L
    for n, tid in ids.items():
        m_ref = (o.opt_d.slots[n] if n in o.opt_d.slots else o.opt_g.slots[n])[0]
        err = _err(eng.get_tensor(tid, slot=L.SLOT_ADAM_M), m_ref)
        print("   GANMF f16 first moment %-3s error %.2e of its scale" % (n, err))
        assert err <= 1e-2, (n, err)
    eng.close()


def test_forced_f32_plans_stay_f32(monkeypatch, capfd):
    """A handle created with mfma="f32" (GANMF_FLAG_MFMA_F32) runs the fp32 MFMA EVERYWHERE: the 16-wave split-bf16 kernel
    (GANMF_X3KG, default on) must not take over its plans, neither the stand-alone products nor the GEMM halves of the combined
    launches.  GANMF_DEBUG_PLAN prints the arithmetic actually launched; the results equal a forced-fp32 handle with the
    16-wave kernel switched off, bit for bit -- and differ from the default arithmetic."""
    from ganmf_amd.engine import Engine
    from ganmf_amd.synthetic import glorot_params, synthetic_urm
    U, N, k, e, B = 300, 3706, 250, 992, 128
    hp = dict(d_lr=1e-4, g_lr=2e-4, d_reg=1e-4, g_reg=0.0, m=10.0, recon_coefficient=0.05)
    urm = synthetic_urm(U, N, 0.04, seed=3)
    w = glorot_params(U, N, k, e, seed=4)
    ids = {"We": 0, "be": 1, "Wd": 2, "bd": 3, "U": 100, "V": 101}
    perm = np.random.RandomState(1).permutation(U)

    def run(mfma, x3kg):
        monkeypatch.setenv("GANMF_DEBUG_PLAN", "1")
        monkeypatch.setenv("GANMF_X3KG", x3kg)
        eng = Engine(U, N, k, e, B, mfma=mfma, **hp)
        eng.set_urm(urm)
        for n, tid in ids.items():
            eng.set_tensor(tid, w[n])
        eng.train_epoch(perm, 1, 1)
        out = {n: eng.get_tensor(tid).copy() for n, tid in ids.items()}
        eng.close()
        plans = [l for l in capfd.readouterr().err.splitlines() if l.startswith("[ganmf plan]")]
        return out, plans

    forced, plans = run("f32", "7")
    assert plans and all("mfma f32" in l for l in plans), [l for l in plans if "mfma f32" not in l]
    ring_only, _ = run("f32", "0")
    for n in ids:
        np.testing.assert_array_equal(forced[n], ring_only[n])
    auto, plans_auto = run(None, "7")
    assert any("mfma bf16x3" in l for l in plans_auto)
    assert any(not np.array_equal(forced[n], auto[n]) for n in ids)
