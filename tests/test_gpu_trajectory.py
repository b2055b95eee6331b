"""Trajectories of 200 updates at BASELINE shapes against the fp64 oracle, with a stated normalised tolerance per horizon.

Per-step agreement (1-6 updates, tests/test_gpu_parity.py, test_gpu_disganmf.py, test_gpu_configs.py) pins the arithmetic;
the end-to-end statistical tests pin where 10^4 updates land.  This file covers the stretch in between: four blocks of
25 discriminator + 25 generator updates (the reference's pass structure, GANMF.py:175-203, over 25 slices of 128 rows
of a fresh permutation per block) at the configs[1] shape (GANMF, ML-1M: 6040 x 3706, k = 250, emb_dim = 992) and the
configs[4] shape (DisGANMF, ML-1M, k = 250, d_nodes = 1024, one linear layer; fp32-accurate default arithmetic and the
fp16 MFMA mode), every tensor and the scores of 257 users compared with the fp64 oracle after 50 / 100 / 150 / 200 updates as
    max |got - ref| / max |ref|      (normalised by the tensor's scale).
What is required at each horizon:
  * GANMF: 1e-5 on every tensor, the scores and every loss of the block (measured on MI355X: 4e-7 .. 1.4e-6, flat from 50 to 200
    updates -- an order of magnitude inside the north star's 1e-4);
  * DisGANMF: fp32 arithmetic ITSELF does not follow the fp64 trajectory of this model to 1e-4 -- the raw float(uid) input
    column (values to 6039) saturates the sigmoid within the first updates, the generator's gradients fall to ~1e-8 of the
    parameters' scale and TF-Adam divides by their own magnitude, so one fp32 rounding of a near-zero gradient moves an element of
    user_embeddings by a learning rate (measured after 50 updates: 2e-3 of max|U| for the HIP path AND for the numpy oracle run
    in float32).  The stated tolerance is therefore relative to that: the same numpy oracle is run in float32 next to the fp64
    one, and the HIP path may deviate from fp64 by at most DIS_FACTOR x what the float32 numpy run deviates (or DIS_FLOOR).
    The fp16 MFMA mode (configs[4] as written) is held on its losses.
The measured values are printed (and recorded in DESIGN.md section 2)."""
import numpy as np
import pytest

from ganmf_amd.synthetic import glorot_params, synthetic_urm
from oracle.ganmf_oracle import DisGANMFOracle, GANMFOracle

pytestmark = [pytest.mark.gpu, pytest.mark.slow]

BLOCKS, SLICES, B = 4, 25, 128


def _err(got, ref):
    return np.max(np.abs(np.asarray(got, np.float64).reshape(np.shape(ref)) - ref)) / (np.max(np.abs(ref)) + 1e-30)


# normalised tolerance after 50 / 100 / 150 / 200 updates
TOL_GANMF = {50: 1e-5, 100: 1e-5, 150: 1e-5, 200: 1e-5}            # north star: scores within 1e-4 rel
DIS_FACTOR, DIS_FLOOR = 4.0, 5e-5                                   # x the float32 numpy oracle's own deviation from fp64
TOL_DIS_F16_LOSS = 2e-3                                             # mixed precision: losses only (operands rounded to 11 bits)


def test_c2_ganmf_200_updates_vs_fp64_oracle():
    from ganmf_amd.engine import Engine
    U, N, k, e = 6040, 3706, 250, 992
    hp = dict(d_lr=1e-4, g_lr=1.6532e-4, d_reg=1e-4, g_reg=0.0, m=10.0, recon_coefficient=0.01)
    urm = synthetic_urm(U, N, 0.035, seed=21)
    w = glorot_params(U, N, k, e, seed=6)
    o = GANMFOracle(U, N, k, e, dtype=np.float64, **hp)
    o.set_params(**w)
    eng = Engine(U, N, k, e, B, **hp)
    eng.set_urm(urm)
    ids = {"We": 0, "be": 1, "Wd": 2, "bd": 3, "U": 100, "V": 101}
    for n, tid in ids.items():
        eng.set_tensor(tid, w[n])
    rng = np.random.RandomState(8)
    probe = rng.permutation(U)[:257]
    for blk in range(1, BLOCKS + 1):
        perm = rng.permutation(U)[:SLICES * B]
        dl_ref, gl_ref = o.train_epoch(urm, perm, B)
        dl, gl = eng.train_epoch(perm, 1, 1)
        T = 2 * SLICES * blk
        errs = {n: _err(eng.get_tensor(tid), o.p[n]) for n, tid in ids.items()}
        errs["scores"] = _err(eng.scores(probe), o.scores(probe))
        errs["dloss"] = float(np.max(np.abs(dl - dl_ref) / np.abs(dl_ref)))
        errs["gloss"] = float(np.max(np.abs(gl - gl_ref) / np.abs(gl_ref)))
        print("C2 GANMF T=%3d: " % T + "  ".join("%s %.2e" % kv for kv in errs.items()))
        for n, v in errs.items():
            assert v <= TOL_GANMF[T], ("C2 GANMF", T, n, v)
    eng.close()


def test_c5_disganmf_200_updates_vs_fp64_oracle():
    """Both arithmetics of configs[4] against ONE fp64 oracle trajectory (and one float32 numpy trajectory, run beside it in a
    second thread): the fp32-accurate default on every tensor, the fp16 MFMA mode on its losses."""
    from concurrent.futures import ThreadPoolExecutor
    from ganmf_amd import _lib as L
    from ganmf_amd.engine import Engine
    U, N, k, e = 6040, 3706, 250, 1024
    hp = dict(d_lr=1e-4, g_lr=5.665e-4, d_reg=3.002e-5, g_reg=0.0, recon_coefficient=0.5)
    urm = synthetic_urm(U, N, 0.035, seed=22)
    o = DisGANMFOracle(U, N, k, d_layers=1, d_nodes=e, d_hidden_act="linear", dtype=np.float64, seed=7, **hp)
    o32 = DisGANMFOracle(U, N, k, d_layers=1, d_nodes=e, d_hidden_act="linear", dtype=np.float32, seed=7, **hp)
    ids = {"W0": 0, "b0": 1, "Wo": 2, "bo": 3, "U": 100, "V": 101}
    engines = {}
    for mfma in (None, "f16"):
        eng = Engine(U, N, k, e, B, model=L.MODEL_DISGANMF, d_layers=1, d_act="linear", m=0.0, mfma=mfma, **hp)
        eng.set_urm(urm)
        for n, tid in ids.items():
            eng.set_tensor(tid, o.p[n])
        engines[mfma] = eng
    rng = np.random.RandomState(9)
    probe = rng.permutation(U)[:257]
    pool = ThreadPoolExecutor(max_workers=2)
    for blk in range(1, BLOCKS + 1):
        perm = rng.permutation(U)[:SLICES * B]
        f64, f32 = pool.submit(o.train_epoch, urm, perm, B), pool.submit(o32.train_epoch, urm, perm, B)
        (dl_ref, gl_ref), (dl32, gl32) = f64.result(), f32.result()
        T = 2 * SLICES * blk
        base = {n: _err(o32.p[n], o.p[n]) for n in ids}
        base["scores"] = _err(o32.scores(probe), o.scores(probe))
        base["dloss"] = float(np.max(np.abs(dl32 - dl_ref) / (np.abs(dl_ref) + 1e-5)))
        base["gloss"] = float(np.max(np.abs(gl32 - gl_ref) / (np.abs(gl_ref) + 1e-5)))
        print("   numpy float32 oracle T=%3d: " % T + "  ".join("%s %.2e" % kv for kv in base.items()))
        for mfma, eng in engines.items():
            dl, gl = eng.train_epoch(perm, 1, 1)
            errs = {n: _err(eng.get_tensor(tid), o.p[n]) for n, tid in ids.items()}
            errs["scores"] = _err(eng.scores(probe), o.scores(probe))
            errs["dloss"] = float(np.max(np.abs(dl - dl_ref) / (np.abs(dl_ref) + 1e-5)))
            errs["gloss"] = float(np.max(np.abs(gl - gl_ref) / (np.abs(gl_ref) + 1e-5)))
            print("C5 DisGANMF %s T=%3d: " % (mfma or "f32-accurate", T) + "  ".join("%s %.2e" % kv for kv in errs.items()))
            if mfma is None:
                for n, v in errs.items():
                    assert v <= max(DIS_FACTOR * base[n], DIS_FLOOR), ("C5 DisGANMF", T, n, v, base[n])
            else:
                assert errs["dloss"] <= TOL_DIS_F16_LOSS and errs["gloss"] <= TOL_DIS_F16_LOSS, (T, errs)
    pool.shutdown()
    for eng in engines.values():
        eng.close()
