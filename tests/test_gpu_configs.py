"""BASELINE.json configs at their full shapes (C1 LastFM-user, C2 ML-1M-user, C3
hetrec-item, C5 DisGANMF ML-1M) plus the edge cases the reference's data can produce: a few
D+G updates against the fp64 oracle on synthetic matrices of the real shapes / densities, and
size-independent properties (bitwise reproducibility, snapshot idempotence)."""
import numpy as np
import pytest
import scipy.sparse as sps

from ganmf_amd.synthetic import glorot_params, synthetic_urm
from oracle.ganmf_oracle import DisGANMFOracle, GANMFOracle

pytestmark = pytest.mark.gpu

NAME2ID = {"We": 0, "be": 1, "Wd": 2, "bd": 3, "U": 100, "V": 101}


def _err(got, ref):
    return np.max(np.abs(np.asarray(got, np.float64).reshape(np.shape(ref)) - ref)) / (np.max(np.abs(ref)) + 1e-30)


CONFIGS = {
    # name: U, N, k, e, B, density, hyper-parameters (tuned values from experiments/*/best_params.txt where they exist)
    "C1_lastfm_user": (1884, 17632, 10, 32, 32, 0.0025, dict(d_lr=1e-4, g_lr=1e-4, d_reg=0.0, g_reg=0.0, m=1.0, recon_coefficient=0.01)),
    "C2_ml1m_user": (6040, 3706, 250, 992, 128, 0.035, dict(d_lr=1e-4, g_lr=1.6532e-4, d_reg=1e-4, g_reg=0.0, m=10.0, recon_coefficient=0.01)),
    # C4 = 200 k x 50 k sharded over 8 GPUs: one rank's shard, 25 k users of the 50 k-wide matrix
    "C4_shard_synthetic": (25000, 50000, 250, 1024, 128, 0.01, dict(d_lr=1e-4, g_lr=1e-4, d_reg=1e-4, g_reg=0.0, m=10.0, recon_coefficient=0.01)),
    "C3_hetrec_item": (10109, 2113, 100, 748, 128, 0.032, dict(d_lr=1e-4, g_lr=1e-4, d_reg=1e-4, g_reg=0.0, m=1.0, recon_coefficient=0.5)),
}


@pytest.mark.parametrize("name", list(CONFIGS))
def test_full_shape_steps_vs_oracle(name):
    from ganmf_amd.engine import Engine
    U, N, k, e, B, dens, hp = CONFIGS[name]
    urm = synthetic_urm(U, N, dens, seed=11)
    w = glorot_params(U, N, k, e, seed=5)
    o = GANMFOracle(U, N, k, e, dtype=np.float64, **hp)
    o.set_params(**w)
    eng = Engine(U, N, k, e, B, **hp)
    eng.set_urm(urm)
    for n, tid in NAME2ID.items():
        eng.set_tensor(tid, w[n])
    rng = np.random.RandomState(3)
    perm = rng.permutation(U)
    for t in range(3):                             # (the fp64 oracle needs ~0.2 TFLOP per update pair at C4)
        uids = perm[t * B:(t + 1) * B]
        X = urm[uids].toarray()
        ld_ref, ld = o.d_step(uids, X), eng.train_step(0, uids)
        lg_ref, lg = o.g_step(uids, X), eng.train_step(1, uids)
        assert abs(ld - ld_ref) <= 5e-5 * abs(ld_ref) + 1e-7, (name, t, ld, ld_ref)
        assert abs(lg - lg_ref) <= 5e-5 * abs(lg_ref) + 1e-7, (name, t, lg, lg_ref)
    for n, tid in NAME2ID.items():
        assert _err(eng.get_tensor(tid), o.p[n]) <= 5e-5, (name, n)
    ids = perm[:257]
    assert _err(eng.scores(ids), o.scores(ids)) <= 1e-4
    eng.close()


def test_c5_disganmf_ml1m_shape():
    from ganmf_amd import _lib as L
    from ganmf_amd.engine import Engine
    U, N, k, e, B = 6040, 3706, 250, 1024, 128
    hp = dict(d_lr=1e-4, g_lr=5.665e-4, d_reg=3.002e-5, g_reg=0.0, recon_coefficient=0.5)
    urm = synthetic_urm(U, N, 0.035, seed=12)
    o = DisGANMFOracle(U, N, k, d_layers=1, d_nodes=e, d_hidden_act="linear", dtype=np.float64, seed=7, **hp)
    eng = Engine(U, N, k, e, B, model=L.MODEL_DISGANMF, d_layers=1, d_act="linear", m=0.0, **hp)
    eng.set_urm(urm)
    ids = {"W0": 0, "b0": 1, "Wo": 2, "bo": 3, "U": 100, "V": 101}
    for n, tid in ids.items():
        eng.set_tensor(tid, o.p[n])
    perm = np.random.RandomState(1).permutation(U)
    for t in range(2):
        uids = perm[t * B:(t + 1) * B]
        X = urm[uids].toarray()
        ld_ref, ld = o.d_step(uids, X), eng.train_step(0, uids)
        lg_ref, lg = o.g_step(uids, X), eng.train_step(1, uids)
        # raw float(uid) up to 6039 multiplies a Glorot weight: logits are O(10..100) and the loss is the
        # cross-entropy of saturated sigmoids; compare with a tolerance scaled to the logit magnitude
        assert abs(ld - ld_ref) <= 2e-4 * abs(ld_ref) + 1e-5, (t, ld, ld_ref)
        assert abs(lg - lg_ref) <= 2e-4 * abs(lg_ref) + 1e-5, (t, lg, lg_ref)
    for n, tid in ids.items():
        assert _err(eng.get_tensor(tid), o.p[n]) <= 1e-4, n
    eng.close()


def test_bitwise_reproducible_and_snapshot_idempotent():
    """Two handles fed the same inputs end bit-identical (no float atomics anywhere);
    snapshot -> train -> restore returns every tensor and moment-independent score."""
    from ganmf_amd.engine import Engine
    U, N, k, e, B = 1500, 2000, 64, 200, 128
    hp = dict(d_lr=1e-3, g_lr=1e-3, d_reg=1e-4, g_reg=1e-5, m=5.0, recon_coefficient=0.1)
    urm = synthetic_urm(U, N, 0.03, seed=2)
    w = glorot_params(U, N, k, e, seed=2)
    engs = []
    for _ in range(2):
        eng = Engine(U, N, k, e, B, **hp)
        eng.set_urm(urm)
        for n, tid in NAME2ID.items():
            eng.set_tensor(tid, w[n])
        engs.append(eng)
    perm = np.random.RandomState(9).permutation(U)
    outs = [eng.train_epoch(perm, 1, 1) for eng in engs]
    np.testing.assert_array_equal(outs[0][0], outs[1][0])
    np.testing.assert_array_equal(outs[0][1], outs[1][1])
    for tid in NAME2ID.values():
        np.testing.assert_array_equal(engs[0].get_tensor(tid), engs[1].get_tensor(tid))
    a = engs[0]
    a.snapshot_best()
    ref = {tid: a.get_tensor(tid) for tid in NAME2ID.values()}
    a.train_epoch(perm, 1, 1)
    a.restore_best()
    for tid in NAME2ID.values():
        np.testing.assert_array_equal(a.get_tensor(tid), ref[tid])
    for eng in engs:
        eng.close()


@pytest.mark.parametrize("U,N,k,e,B", [(5, 63, 1, 1, 8), (130, 64, 3, 63, 200), (64, 127, 64, 64, 64), (33, 191, 7, 127, 5)])
def test_edge_shapes(U, N, k, e, B):
    """dims of 1, batch larger than the matrix, widths that put the ones / uid columns on a
    64-float boundary, empty profile rows (cold users are legal CSR rows)."""
    from ganmf_amd.engine import Engine
    rng = np.random.RandomState(U + N)
    m = (rng.rand(U, N) < 0.1).astype(np.float32)
    m[::3] = 0.0                                   # cold rows
    urm = sps.csr_matrix(m)
    hp = dict(d_lr=1e-3, g_lr=1e-3, d_reg=1e-3, g_reg=1e-4, m=2.0, recon_coefficient=0.2)
    o = GANMFOracle(U, N, k, e, dtype=np.float64, seed=1, **hp)
    eng = Engine(U, N, k, e, B, **hp)
    eng.set_urm(urm)
    for n, tid in NAME2ID.items():
        eng.set_tensor(tid, o.p[n])
    perm = rng.permutation(U)
    for _ in range(2):
        dl_ref, gl_ref = o.train_epoch(urm, perm, min(B, U), 1, 1)
        dl, gl = eng.train_epoch(perm, 1, 1)
        np.testing.assert_allclose(dl, dl_ref, rtol=1e-4, atol=1e-7)
        np.testing.assert_allclose(gl, gl_ref, rtol=1e-4, atol=1e-7)
    for n, tid in NAME2ID.items():
        assert _err(eng.get_tensor(tid), o.p[n]) <= 1e-4, n
    eng.close()


@pytest.mark.parametrize("k,e,B", [(10, 32, 32), (16, 100, 256), (67, 398, 1024)])
def test_c1_sparse_real_path(golden_dir, monkeypatch, k, e, B):
    """SURVEY 8(f)-3 at BASELINE configs[0] (LastFM 1884 x 17632, 0.22 % dense; the reference's default k / emb_dim / batch,
    a wider one, and the reference's TUNED k = 67, emb_dim = 398, batch = 1024): the real rows X stay CSR.
      generator step      Er = X.We + be as a CSR row-sum, X never expanded                              (GANMF.py:198-201)
      discriminator step  the same for Er; the residual R - X takes X from a CSR lookup in the decode epilogue (bit-identical to
                          the dense subtraction); the encoder gradient's real half X^T.dE_r comes from the CSC matrix in the
                          epilogue of a GEMM that runs over the generated rows only                       (GANMF.py:183-187)
    One epoch (ragged last batch) in every combination of the two paths and with the planner's own choice against the fp64
    oracle; the planner must pick the sparse generator path at this density, and the sparse discriminator path exactly where
    it removes >= 2 GFLOP per step (the tuned configuration).  Two handles of the sparse path end bit-identical (fixed summation
    order: no float atomics)."""
    import os
    from ganmf_amd.engine import Engine
    urm = sps.load_npz(os.path.join(golden_dir, "LastFM_URM_train.npz")).tocsr().astype(np.float32)
    U, N = urm.shape
    assert urm.nnz / (U * N) < 0.005
    hp = dict(d_lr=1e-3, g_lr=1e-3, d_reg=1e-4, g_reg=1e-5, m=5.0, recon_coefficient=0.3)
    o = GANMFOracle(U, N, k, e, dtype=np.float64, seed=3, **hp)
    p0 = o.get_params()
    perm = np.random.RandomState(4).permutation(U)
    dl_ref, gl_ref = o.train_epoch(urm, perm, B, 1, 1)
    got = {}
    for mode in (("1", "1"), ("1", "1"), ("1", "0"), ("0", None), (None, None)):
        for var, val in zip(("GANMF_SPARSE", "GANMF_SPARSE_D"), mode):
            if val is None:
                monkeypatch.delenv(var, raising=False)
            else:
                monkeypatch.setenv(var, val)
        eng = Engine(U, N, k, e, B, **hp)
        eng.set_urm(urm)
        for n, tid in NAME2ID.items():
            eng.set_tensor(tid, p0[n])
        dl, gl = eng.train_epoch(perm, 1, 1)
        np.testing.assert_allclose(dl, dl_ref, rtol=1e-4, atol=1e-7)
        np.testing.assert_allclose(gl, gl_ref, rtol=1e-4, atol=1e-7)
        res = {n: eng.get_tensor(tid) for n, tid in NAME2ID.items()}
        for n in NAME2ID:
            assert _err(res[n], o.p[n]) <= 1e-4, (mode, n)
        if mode in got:      # second handle of the fully sparse path: bitwise reproducible
            for n in NAME2ID:
                assert np.array_equal(got[mode][n], res[n]), n
        got[mode] = res
        eng.close()
    expect = ("1", "1") if 4.0 * min(B, U) * N * e >= 2.0e9 else ("1", "0")
    for n in NAME2ID:
        assert np.array_equal(got[(None, None)][n], got[expect][n]), (n, expect)          # the planner's choice
    assert any(not np.array_equal(got[("0", None)][n], got[("1", "0")][n]) for n in NAME2ID)      # ... is a different summation order
    assert any(not np.array_equal(got[("1", "1")][n], got[("1", "0")][n]) for n in NAME2ID)
