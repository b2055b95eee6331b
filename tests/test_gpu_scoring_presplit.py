"""The scoring product U[ids] . V^T (GANMF.py:285-292) on the pre-split persistent kernel (csrc/gemm_bf16p.hpp): many-tile shapes
under the default fp32-accurate arithmetic split both factors ONCE into three bf16 planes and run one persistent launch.

  * accuracy against float64 at the fp32 bound, element by element relative to sum |a||b| (the bound the fp32 MFMA path and the
    one-tile split-bf16 kernel are held to in test_gpu_mfma_modes.py), at the edges of the kernel: K of a single K-tile, K beyond
    the eight K-steps a tile's stores ride under, ragged M and N, both orientations, rows gathered through an id list;
  * the other factor's planes are cached between calls: they must follow every change of the parameters (set_tensor, training,
    restore_best)."""
import numpy as np
import pytest
import scipy.sparse as sps

pytestmark = pytest.mark.gpu


def _factors(rng, U, N, k):
    Uf = (rng.standard_normal((U, k)) * np.exp(rng.uniform(-3, 3, (U, 1)))).astype(np.float32)
    Vf = (rng.standard_normal((N, k)) * np.exp(rng.uniform(-3, 3, (N, 1)))).astype(np.float32)
    return Uf, Vf


@pytest.mark.parametrize("U,N,k", [(6040, 3706, 250), (4001, 4100, 7), (3000, 9000, 64), (5000, 5000, 300), (6040, 3706, 33),
                                    (1884, 17632, 1)])
def test_presplit_scores_at_fp32_accuracy(monkeypatch, U, N, k):
    from ganmf_amd.engine import Engine
    rng = np.random.RandomState(U + k)
    Uf, Vf = _factors(rng, U, N, k)
    ref = Uf.astype(np.float64) @ Vf.astype(np.float64).T
    bound = np.abs(Uf).astype(np.float64) @ np.abs(Vf).astype(np.float64).T + 1e-300
    ids = rng.permutation(U)[: U - 3]
    idt = np.arange(N - 5)
    got = {}
    monkeypatch.setenv("GANMF_TUNE", "skinny=0")      # (K <= 64 would otherwise stream through gemm_skinny.hpp: its own test in test_gpu_gemm.py)
    for pre in ("1", "0"):
        monkeypatch.setenv("GANMF_SCORE_PRESPLIT", pre)
        eng = Engine(U, N, k, 8, 8)
        eng.set_tensor(100, Uf)
        eng.set_tensor(101, Vf)
        s, st = eng.scores(ids), eng.scores(idt, transposed=True)
        assert np.max(np.abs(s - ref[ids]) / bound[ids]) < 4e-7, (pre, "user mode")
        assert np.max(np.abs(st - ref.T[idt]) / bound.T[idt]) < 4e-7, (pre, "item mode")
        got[pre] = s
        eng.close()
    assert np.max(np.abs(got["1"] - got["0"]) / bound[ids]) < 4e-7      # two summation orders of the same six piece products


def test_cached_planes_follow_the_parameters():
    from ganmf_amd.engine import Engine
    rng = np.random.RandomState(1)
    U, N, k, e, B = 4096, 3200, 40, 16, 64
    urm = sps.random(U, N, density=0.01, format="csr", random_state=2, dtype=np.float32)
    urm.data[:] = 1.0
    eng = Engine(U, N, k, e, B, d_lr=1e-3, g_lr=1e-2, d_reg=1e-4, m=5.0, recon_coefficient=0.2)
    eng.set_urm(urm)
    w = {0: rng.randn(N, e) * 0.05, 1: np.zeros(e), 2: rng.randn(e, N) * 0.05, 3: np.zeros(N),
         100: rng.randn(U, k) * 0.1, 101: rng.randn(N, k) * 0.1}
    for tid, a in w.items():
        eng.set_tensor(tid, a.astype(np.float32))
    ids = np.arange(U)

    def check(what):
        Uc, Vc = eng.get_tensor(100).astype(np.float64), eng.get_tensor(101).astype(np.float64)
        s = eng.scores(ids)
        bound = np.abs(Uc) @ np.abs(Vc).T + 1e-300
        assert np.max(np.abs(s - Uc @ Vc.T) / bound) < 4e-7, what
        return s
    s0 = check("initial")
    np.testing.assert_array_equal(eng.scores(ids), s0)                 # same parameters: cached planes, same result
    eng.set_tensor(101, (rng.randn(N, k) * 0.1).astype(np.float32))
    assert not np.array_equal(check("after set_tensor(V)"), s0)
    eng.snapshot_best()
    s1 = eng.scores(ids)
    eng.train_epoch(rng.permutation(U)[: 4 * B], 1, 1)                 # four D and four G updates: V moves (and swaps buffers)
    assert not np.array_equal(check("after training"), s1)
    eng.restore_best()
    np.testing.assert_array_equal(check("after restore_best"), s1)
    st = eng.scores(np.arange(N), transposed=True)                     # the other orientation uses the other factor's planes
    np.testing.assert_allclose(st, s1.T, rtol=0, atol=4e-7 * float(np.max(np.abs(s1))))
    eng.close()
