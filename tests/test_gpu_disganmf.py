"""DisGANMF on the HIP path vs the fp64 oracle, through the C ABI:
every activation, 1 and 2 hidden layers, single steps, epochs with ragged tails, the committed
golden trajectory.  Tolerances as in test_gpu_parity.py (normalised by the tensor's scale)."""
import json
import os

import numpy as np
import pytest
import scipy.sparse as sps

from oracle.ganmf_oracle import ACTIVATIONS, DisGANMFOracle, reference_epoch_permutations

pytestmark = pytest.mark.gpu


def _rand_urm(rng, U, N, density):
    m = (rng.rand(U, N) < density).astype(np.float32)
    m[np.arange(U), rng.randint(0, N, U)] = 1.0
    return sps.csr_matrix(m)


def _ids(L):
    d = {}
    for l in range(L):
        d["W%d" % l] = 2 * l
        d["b%d" % l] = 2 * l + 1
    d["Wo"], d["bo"], d["U"], d["V"] = 2 * L, 2 * L + 1, 100, 101
    return d


def _engine(o, urm, B, hp, L, e, act):
    from ganmf_amd import _lib as LL
    from ganmf_amd.engine import Engine
    eng = Engine(o.nu, o.ni, o.k, e, B, model=LL.MODEL_DISGANMF, d_layers=L, d_act=act, m=0.0, **hp)
    eng.set_urm(urm)
    for n, tid in _ids(L).items():
        eng.set_tensor(tid, o.p[n])
    return eng


def _close(got, ref, rtol, what):
    got = np.asarray(got).reshape(np.shape(ref))
    scale = np.max(np.abs(ref)) + 1e-30
    err = np.max(np.abs(got.astype(np.float64) - np.asarray(ref, dtype=np.float64))) / scale
    assert err <= rtol, "%s: normalised error %.3e > %.1e" % (what, err, rtol)


HP = dict(d_lr=1e-3, g_lr=2e-3, d_reg=1e-4, g_reg=0.0, recon_coefficient=0.3)


@pytest.mark.parametrize("act", ACTIVATIONS)
@pytest.mark.parametrize("layers", [1, 2])
@pytest.mark.parametrize("shape", [(41, 67, 5, 9, 8), (300, 517, 33, 70, 64)])
def test_disganmf_single_steps(act, layers, shape):
    U, N, k, e, B = shape
    rng = np.random.RandomState(U + layers)
    urm = _rand_urm(rng, U, N, 0.06)
    o = DisGANMFOracle(U, N, k, d_layers=layers, d_nodes=e, d_hidden_act=act, dtype=np.float64, seed=3, **HP)
    # shrink the uid row so that float(uid) * w does not saturate tanh/sigmoid at the very first step
    o.p["W0"][0, :] *= 1.0 / U
    for l in range(layers):
        o.p["b%d" % l] = rng.randn(e) * 0.01
    eng = _engine(o, urm, B, HP, layers, e, act)
    uids = rng.permutation(U)[:B]
    X = urm[uids].toarray()
    ld_ref = o.d_step(uids, X)
    ld = eng.train_step(0, uids)
    assert abs(ld - ld_ref) <= 3e-5 * abs(ld_ref) + 1e-7, (ld, ld_ref)
    for n in o.D_NAMES:
        _close(eng.get_tensor(_ids(layers)[n]), o.p[n], 3e-5, "D-step " + n)
    uids2 = rng.permutation(U)[:max(B - 3, 1)]
    X2 = urm[uids2].toarray()
    lg_ref = o.g_step(uids2, X2)
    lg = eng.train_step(1, uids2)
    assert abs(lg - lg_ref) <= 3e-5 * abs(lg_ref) + 1e-7, (lg, lg_ref)
    for n in ("U", "V"):
        _close(eng.get_tensor(_ids(layers)[n]), o.p[n], 3e-5, "G-step " + n)
    eng.close()


def test_disganmf_epochs_ragged_tanh2():
    U, N, k, e, B, L = 101, 160, 12, 20, 16, 2
    rng = np.random.RandomState(8)
    urm = _rand_urm(rng, U, N, 0.07)
    o = DisGANMFOracle(U, N, k, d_layers=L, d_nodes=e, d_hidden_act="tanh", dtype=np.float64, seed=2, **HP)
    o.p["W0"][0, :] *= 1.0 / U
    eng = _engine(o, urm, B, HP, L, e, "tanh")
    for perm in reference_epoch_permutations(U, 3, 1337):
        dl_ref, gl_ref = o.train_epoch(urm, perm, B, 1, 1)
        dl, gl = eng.train_epoch(perm, 1, 1)
        np.testing.assert_allclose(dl, dl_ref, rtol=1e-4, atol=1e-7)
        np.testing.assert_allclose(gl, gl_ref, rtol=1e-4, atol=1e-7)
    for n, tid in _ids(L).items():
        _close(eng.get_tensor(tid), o.p[n], 2e-4, "epochs " + n)
    eng.close()


def test_disganmf_golden_trajectory_and_class(golden_dir):
    g = np.load(os.path.join(golden_dir, "tiny_trajectories.npz"))
    urm = sps.load_npz(os.path.join(golden_dir, "tiny_urm.npz")).tocsr()
    name = "disganmf_user_tanh2"
    c = json.loads(str(g[name + "/config"]))
    from GANRec.DisGANMF import DisGANMF
    assert DisGANMF.__module__.split(".")[0] == "GANRec"
    np.random.seed(1337)
    model = DisGANMF(urm, mode='user', seed=11, is_experiment=True)
    L = c["hp"]["d_layers"]
    model.initial_weights = {n: g["%s/init/%s" % (name, n)] for n in _ids(L)}
    hp = dict(c["hp"])
    ret = model.fit(num_factors=c["k"], d_nodes=c["e"], epochs=c["epochs"], batch_size=c["B"], d_steps=c["d_steps"],
                    g_steps=c["g_steps"], **hp)
    assert ret == c["epochs"] + 1
    for n, tid in _ids(L).items():
        _close(model.engine.get_tensor(tid), g["%s/f64/final/%s" % (name, n)], 2e-4, "golden " + n)
    np.testing.assert_allclose(model.train_d_loss[-1], np.mean(g[name + "/f64/dloss"][-5:]), rtol=1e-3)
    ids = np.array([3, 0, 36])
    ref = g[name + "/f64/final/U"][ids] @ g[name + "/f64/final/V"].T
    _close(model._compute_item_score(ids), ref, 2e-4, "scores")
    assert [r.name for r in model.params['D']][:2] == ['discriminator/layer_0/kernel', 'discriminator/layer_0/bias']
    assert model.sess.run(model.params['D'][0]).shape == (urm.shape[1] + 1, c["e"])


@pytest.mark.parametrize("layers,act", [(1, "linear"), (2, "tanh")])
def test_head_inside_the_slab_sum_launch_matches_the_separate_head(layers, act, monkeypatch):
    """configs[4]-sized DisGANMF: the last hidden layer's forward GEMM is split along K, so its slab sum runs as
    reduce_rows_head_kernel (logit, cross-entropy and dlogit of a row formed by the workgroup that sums the row).  Against the
    separate dis_head_kernel (GANMF_TUNE=dis_head_fuse=0): the layer outputs are the same numbers, only the order of the 1 025
    products of a row's logit differs -- every tensor and loss after an epoch agrees to fp32 rounding, and both agree with the
    fp64 oracle as test_disganmf_epochs demands."""
    U, N, k, e, B = 520, 3706, 250, 1024, 128
    rng = np.random.RandomState(17)
    urm = _rand_urm(rng, U, N, 0.04)
    hp = dict(d_lr=1e-4, g_lr=5e-4, d_reg=3e-5, g_reg=0.0, recon_coefficient=0.5)
    o = DisGANMFOracle(U, N, k, d_layers=layers, d_nodes=e, d_hidden_act=act, dtype=np.float64, seed=5, **hp)
    o.p["W0"][0, :] *= 1.0 / U
    perm = rng.permutation(U)
    outs = {}
    # "1": head in the slab sum and (one layer, round 6) dz_0 rows written there too, the column sums of the backward top inside the gradient
    # GEMM's launch (gemm_bf16s_colsum); "1,dis_top_gw=0": head in the slab sum, dis_dz_top_kernel as its own launch; "0": every piece its own kernel
    for fuse in ("1", "1,dis_top_gw=0", "0"):
        monkeypatch.setenv("GANMF_TUNE", "dis_head_fuse=" + fuse)
        eng = _engine(o, urm, B, hp, layers, e, act)
        dl, gl = eng.train_epoch(perm, 1, 1)
        outs[fuse] = ({n: eng.get_tensor(tid).copy() for n, tid in _ids(layers).items()}, np.array(dl), np.array(gl))
        eng.close()
    # both forms against the fp64 oracle at the per-epoch bound of this file; against each other within twice that (TF-Adam turns the
    # last bit of a logit into a visible change of a bias that has barely left zero: b1 differs by 8e-5 of its own 4e-4 scale)
    dl_ref, gl_ref = o.train_epoch(urm, perm, B)
    for fuse in ("1", "1,dis_top_gw=0", "0"):
        for n in outs[fuse][0]:
            _close(outs[fuse][0][n], o.p[n], 1e-4, "head fused=%s vs fp64 oracle: %s" % (fuse, n))
        np.testing.assert_allclose(outs[fuse][1], dl_ref, rtol=2e-4)
        np.testing.assert_allclose(outs[fuse][2], gl_ref, rtol=2e-4)
    for other in ("1,dis_top_gw=0", "0"):
        for n in outs["1"][0]:
            _close(outs["1"][0][n], outs[other][0][n], 2e-4, "fused vs %s: %s" % (other, n))
        np.testing.assert_allclose(outs["1"][1], outs[other][1], rtol=2e-5)
        np.testing.assert_allclose(outs["1"][2], outs[other][2], rtol=2e-5)
