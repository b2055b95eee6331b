"""The row-sharded fit() (north star: "users shard row-wise across the 8 GPUs of one node"; reference site of the loop
being sharded: GANRec/GANMF.py:155-203, of the scoring that needs the gathered rows: :285-292) through the host classes'
own entry point -- GANMF(..., dist_backend=..., world_size=...).fit(...) -- on ONE GPU:

  * N ranks on the library's loopback communicator (world 3 / 4): the sharded fit replays the reference's minibatch
    schedule split by row owner, so it must follow the single-GPU fit AND the fp64 oracle's epochs at the per-step
    tolerances; scores, recommend and the hold-out evaluator work on the gathered factors; early stopping (snapshot /
    restore through the master engine) stops at the same epoch;
  * ML-1M with world 4: the same +-0.005 MAP@5 band around the published row as the single-GPU statistical KAT;
  * one rank PROCESS per GPU over RCCL, world 1 (all a one-GPU box allows: RCCL takes one rank per device) with
    GANMF_FORCE_COLLECTIVES=1 so that the in-place ncclReduceScatter / ncclAllGather calls execute: bit-equal to the plain
    engine."""
import json
import os

import numpy as np
import pytest
import scipy.sparse as sps

from oracle.ganmf_oracle import DisGANMFOracle, GANMFOracle, reference_epoch_permutations

pytestmark = pytest.mark.gpu


def _err(got, ref):
    return np.max(np.abs(np.asarray(got, np.float64).reshape(np.shape(ref)) - ref)) / (np.max(np.abs(ref)) + 1e-30)


def _urm(rng, U, N, density):
    m = (rng.rand(U, N) < density).astype(np.float32)
    m[np.arange(U), rng.randint(0, N, U)] = 1.0
    return sps.csr_matrix(m)


@pytest.mark.parametrize("mode,world", [("user", 3), ("item", 4), ("user", 8)])
def test_sharded_ganmf_fit_follows_single_gpu_fit_and_oracle(mode, world):
    from ganmf_amd.GANMF import GANMF
    rng = np.random.RandomState(world)
    urm = _urm(rng, 83, 131, 0.12)
    fit_rows, fit_cols = (urm.shape if mode == "user" else urm.shape[::-1])
    k, e, B, epochs = 6, 11, 16, 3
    hp = dict(d_lr=1e-3, g_lr=2e-3, d_reg=1e-3, g_reg=1e-4, m=10.0, recon_coefficient=0.2)
    o = GANMFOracle(fit_rows, fit_cols, k, e, dtype=np.float64, seed=3, **hp)
    o.set_params(be=rng.randn(e) * 0.01, bd=rng.randn(fit_cols) * 0.01)
    w0 = o.get_params()
    models = {}
    for name, kw in (("single", {}), ("sharded", dict(dist_backend="local", world_size=world))):
        np.random.seed(77)
        m = GANMF(urm, mode=mode, seed=1, is_experiment=True, **kw)
        m.initial_weights = w0
        assert m.fit(num_factors=k, emb_dim=e, epochs=epochs, batch_size=B, **hp) == epochs + 1
        models[name] = m
    assert type(models["sharded"].engine).__name__ == "ShardedEngine" and models["sharded"].engine.world == world
    fit_urm = urm if mode == "user" else urm.T.tocsr()
    for perm in reference_epoch_permutations(fit_rows, epochs, 77):
        dl_ref, gl_ref = o.train_epoch(fit_urm, perm, B)
    np.testing.assert_allclose(models["sharded"].train_d_loss[-1], np.mean(dl_ref), rtol=1e-4)
    np.testing.assert_allclose(models["sharded"].train_g_loss[-1], np.mean(gl_ref), rtol=1e-4)
    for n, tid in (("We", 0), ("be", 1), ("Wd", 2), ("bd", 3), ("U", 100), ("V", 101)):
        got = models["sharded"].engine.get_tensor(tid)
        assert _err(got, o.p[n]) <= 1e-4, n
        assert _err(got, models["single"].engine.get_tensor(tid).astype(np.float64)) <= 1e-4, n
    ids = np.arange(urm.shape[0])
    s = models["sharded"]._compute_item_score(ids)
    assert _err(s, o.scores(ids, item_mode=(mode == "item"))) <= 1e-4
    # recommend() / the device evaluator run on the gathered factors
    rec_a = models["sharded"].recommend(ids[:20], cutoff=5)
    rec_b = models["single"].recommend(ids[:20], cutoff=5)
    masked = np.where(urm[ids[:20]].toarray() > 0, -np.inf, s[:20])
    for u, (a, b) in enumerate(zip(rec_a, rec_b)):
        assert len(a) == 5 and np.allclose(np.sort(masked[u])[::-1][:5], masked[u][a], rtol=1e-4, atol=1e-6)
    for m in models.values():
        m.engine.close()


def test_sharded_disganmf_fit_follows_oracle():
    from ganmf_amd.DisGANMF import DisGANMF
    rng = np.random.RandomState(9)
    U, N, k, e, B, epochs, world = 61, 97, 5, 12, 16, 2, 3
    urm = _urm(rng, U, N, 0.15)
    hp = dict(d_lr=1e-3, g_lr=1e-3, d_reg=1e-4, g_reg=1e-5, recon_coefficient=0.3)
    o = DisGANMFOracle(U, N, k, d_layers=2, d_nodes=e, d_hidden_act="tanh", dtype=np.float64, seed=5, **hp)
    w0 = o.get_params()
    np.random.seed(5)
    m = DisGANMF(urm, mode="user", seed=5, is_experiment=True, dist_backend="local", world_size=world)
    m.initial_weights = w0
    m.fit(num_factors=k, d_layers=2, d_nodes=e, d_hidden_act="tanh", epochs=epochs, batch_size=B, **hp)
    for perm in reference_epoch_permutations(U, epochs, 5):
        dl_ref, gl_ref = o.train_epoch(urm, perm, B)
    np.testing.assert_allclose(m.train_d_loss[-1], np.mean(dl_ref), rtol=2e-4)
    np.testing.assert_allclose(m.train_g_loss[-1], np.mean(gl_ref), rtol=2e-4)
    ids = {"W0": 0, "b0": 1, "W1": 2, "b1": 3, "Wo": 4, "bo": 5, "U": 100, "V": 101}
    for n, tid in ids.items():
        assert _err(m.engine.get_tensor(tid), o.p[n]) <= 2e-4, n      # (float(uid) is the GLOBAL row id on every rank)
    m.engine.close()


def test_sharded_fit_early_stopping_matches_single_gpu(golden_dir):
    """fit() with the GAN early-stopping dict (RecSysExp.py:217-223) on a LastFM-sized problem: snapshots and the final
    restore go through the master engine; the sharded run must stop at the epoch the single-GPU run stops at and end on the
    same metrics."""
    from ganmf_amd.GANMF import GANMF
    from ganmf_amd.evaluation import EvaluatorHoldoutFast
    train = sps.load_npz(os.path.join(golden_dir, "LastFM_URM_train.npz")).tocsr()
    test = sps.load_npz(os.path.join(golden_dir, "LastFM_URM_test.npz")).tocsr()
    hp = dict(num_factors=16, emb_dim=64, batch_size=256, d_lr=1e-3, g_lr=5e-3, d_reg=1e-5, m=5, recon_coefficient=0.3)
    out = {}
    for name, kw in (("single", {}), ("sharded", dict(dist_backend="local", world_size=4))):
        np.random.seed(11)
        m = GANMF(train, mode="user", seed=11, is_experiment=True, **kw)
        ev = EvaluatorHoldoutFast(test, [5])
        ret = m.fit(epochs=40, allow_worse=2, freq=2, validation_evaluator=ev, **hp)
        res = EvaluatorHoldoutFast(test, [5]).evaluateRecommender(m)[0][5]
        out[name] = (ret, res["MAP"], res["NDCG"])
        m.engine.close()
    print("early stopping: single %r sharded %r" % (out["single"], out["sharded"]))
    assert out["single"][0] == out["sharded"][0]
    assert abs(out["single"][1] - out["sharded"][1]) <= 0.004 and abs(out["single"][2] - out["sharded"][2]) <= 0.004


def test_sharded_model_saves_and_loads_in_reference_format(tmp_path):
    """saveModel / loadModel (GANMF.py:309-339) on a row-sharded model: the bundle holds the GATHERED user embeddings and loads into a
    sharded model (rows scattered back to their owners: one more epoch continues bit-identically on both) and into an unsharded one."""
    from ganmf_amd.GANMF import GANMF
    rng = np.random.RandomState(5)
    urm = _urm(rng, 90, 70, 0.1)
    kw = dict(dist_backend="local", world_size=3)
    hp = dict(num_factors=5, emb_dim=9, batch_size=16, d_lr=1e-3, g_lr=1e-3, d_reg=1e-4, recon_coefficient=0.1)
    np.random.seed(3)
    a = GANMF(urm, mode="user", seed=4, is_experiment=True, **kw)
    a.fit(epochs=2, **hp)
    ids = np.arange(urm.shape[0])
    before = a._compute_item_score(ids)
    a.saveModel(str(tmp_path))
    b = GANMF(urm, mode="user", is_experiment=True, **kw)
    b.loadModel(str(tmp_path))
    assert type(b.engine).__name__ == "ShardedEngine"
    np.testing.assert_array_equal(b._compute_item_score(ids), before)
    c = GANMF(urm, mode="user", is_experiment=True)
    c.loadModel(str(tmp_path))
    np.testing.assert_array_equal(c._compute_item_score(ids), before)
    for tid in (0, 1, 2, 3, 100, 101):
        np.testing.assert_array_equal(b.engine.get_tensor(tid), a.engine.get_tensor(tid))
        np.testing.assert_array_equal(c.engine.get_tensor(tid), a.engine.get_tensor(tid))
    assert b.recommend(ids[:10], cutoff=5) == a.recommend(ids[:10], cutoff=5)
    for m in (a, b, c):
        m.engine.close()


@pytest.mark.slow
def test_ml1m_sharded_world4_reaches_published_map(golden_dir):
    """BASELINE configs[1] trained through the sharded entry point with four ranks: same band as the single-GPU KAT."""
    from ganmf_amd.GANMF import GANMF
    from ganmf_amd.evaluation import EvaluatorHoldoutFast
    kat = json.load(open(os.path.join(golden_dir, "statistical_kat_ml1m_user.json")))
    train = sps.load_npz(os.path.join(golden_dir, "Movielens1M_URM_train.npz")).tocsr()
    test = sps.load_npz(os.path.join(golden_dir, "Movielens1M_URM_test.npz")).tocsr()
    np.random.seed(1337)
    model = GANMF(train, mode='user', seed=1337, is_experiment=True, dist_backend="local", world_size=4)
    ret = model.fit(validation_set=None, sample_every=None, validation_evaluator=None, **kat["best_params"])
    assert ret == kat["best_params"]["epochs"] + 1
    res, _ = EvaluatorHoldoutFast(test, [5, 10, 20, 50]).evaluateRecommender(model)
    pub = kat["published"]
    print("ML-1M GANMF-user, 4 ranks (loopback): MAP@5 %.4f (published %.4f) NDCG@5 %.4f (%.4f)"
          % (res[5]["MAP"], pub["5"]["MAP"], res[5]["NDCG"], pub["5"]["NDCG"]))
    for metric in ("MAP", "NDCG", "PRECISION", "RECALL"):
        assert abs(res[5][metric] - pub["5"][metric]) <= 0.005, (metric, res[5][metric], pub["5"][metric])
    for c in ("10", "20", "50"):
        assert abs(res[int(c)]["MAP"] - pub[c]["MAP"]) <= 0.005
    model.engine.close()


@pytest.mark.parametrize("e", [32, 1024])
def test_c4_full_shape_sharded_fit_trains_and_scores(e):
    """BASELINE configs[3] through the product entry point: a 200 000 x 50 000 matrix (0.2 % dense), k = 250, eight row shards
    of 25 000 users on the loopback communicator, emb_dim 32 and 1024, one epoch of the reference's schedule at global batch 8 x 128 (196 D + 196 G
    updates), then scores / recommend on the gathered factors -- against the same fit on one unsharded engine."""
    from ganmf_amd.GANMF import GANMF
    from ganmf_amd.synthetic import glorot_params
    U, N, k, B, world, per_row = 200000, 50000, 250, 1024, 8, 100
    rng = np.random.RandomState(11)
    strides = np.array([3, 7, 9, 11, 13, 17, 19, 21, 23, 27, 29, 31, 33, 37, 39, 41])      # coprime with N = 2^4 . 5^5: distinct columns
    cols = (rng.randint(0, N, U)[:, None].astype(np.int64) + strides[rng.randint(0, len(strides), U)][:, None] * np.arange(per_row)[None, :] * 97) % N
    urm = sps.csr_matrix((np.ones(U * per_row, np.float32), cols.astype(np.int32).ravel(), np.arange(0, U * per_row + 1, per_row)), shape=(U, N))
    urm.sort_indices()
    assert urm.nnz == U * per_row and int(np.diff(urm.indptr).min()) == per_row and urm.data.max() == 1.0
    hp = dict(d_lr=1e-4, g_lr=1e-3, d_reg=1e-4, g_reg=0.0, m=1.0, recon_coefficient=0.05)
    w0 = glorot_params(U, N, k, e, seed=9)
    w0["be"] = (rng.randn(e) * 0.01).astype(np.float32)
    models = {}
    for name, kw in (("single", {}), ("sharded", dict(dist_backend="local", world_size=world))):
        np.random.seed(5)
        m = GANMF(urm, mode="user", seed=1, is_experiment=True, **kw)
        m.initial_weights = w0
        assert m.fit(num_factors=k, emb_dim=e, epochs=1, batch_size=B, **hp) == 2
        models[name] = m
    sh, one = models["sharded"], models["single"]
    assert type(sh.engine).__name__ == "ShardedEngine" and sh.engine.world == world
    assert len(sh.train_d_loss) == 1 and np.isfinite(sh.train_d_loss[-1]) and np.isfinite(sh.train_g_loss[-1])
    np.testing.assert_allclose(sh.train_d_loss[-1], one.train_d_loss[-1], rtol=1e-4)
    np.testing.assert_allclose(sh.train_g_loss[-1], one.train_g_loss[-1], rtol=1e-4)
    for n, tid in (("We", 0), ("be", 1), ("Wd", 2), ("bd", 3), ("U", 100), ("V", 101)):
        a, b = sh.engine.get_tensor(tid), one.engine.get_tensor(tid).astype(np.float64)
        assert _err(a, b) <= 1e-4, n
        assert _err(a, np.asarray(w0[n], np.float64).reshape(b.shape)) > 1e-4 or n == "bd", n      # ... and it did train
    ids = np.concatenate([np.arange(world) * (U // world) + r for r in (0, 1, 12345, U // world - 1)])      # every shard, its edges included
    s_sh, s_one = sh._compute_item_score(ids), one._compute_item_score(ids)
    assert s_sh.shape == (len(ids), N) and _err(s_sh, s_one.astype(np.float64)) <= 1e-4
    uf = sh.engine.get_tensor(100)[ids].astype(np.float64) @ sh.engine.get_tensor(101).astype(np.float64).T      # USER_factors[ids] @ ITEM_factors.T
    assert _err(s_sh, uf) <= 1e-5
    rec = sh.recommend(ids[:8], cutoff=10)
    masked = np.where(urm[ids[:8]].toarray() > 0, -np.inf, s_sh[:8])
    for u, r in enumerate(rec):
        assert len(r) == 10 and np.allclose(np.sort(masked[u])[::-1][:10], masked[u][r], rtol=1e-4, atol=1e-7)
    for m in models.values():
        m.engine.close()


def test_process_backend_one_rank_rccl_collectives_execute(monkeypatch):
    """One rank process on the GPU, RCCL communicator, GANMF_FORCE_COLLECTIVES=1 (inherited by the rank process): the
    reduce-scatter / Adam-on-slice / all-gather path with the REAL in-place RCCL calls, ragged owner-split schedule, against
    the plain engine of this process -- bit for bit (same GEMM tiles and arithmetic, one slice = the whole tensor)."""
    from ganmf_amd.dist import ShardedEngine
    from ganmf_amd.engine import Engine
    monkeypatch.setenv("GANMF_FORCE_COLLECTIVES", "1")
    rng = np.random.RandomState(2)
    U, N, k, e, B = 150, 210, 9, 17, 32
    urm = _urm(rng, U, N, 0.08)
    hp = dict(d_lr=1e-3, g_lr=1e-3, d_reg=1e-4, g_reg=1e-5, m=5.0, recon_coefficient=0.1)
    o = GANMFOracle(U, N, k, e, seed=4, **hp)
    sh = ShardedEngine(U, N, k, e, B, devices=[0], backend="process", **hp)
    plain = Engine(U, N, k, e, B, **hp)
    try:
        for eng in (sh, plain):
            eng.set_urm(urm)
            for n, tid in (("We", 0), ("be", 1), ("Wd", 2), ("bd", 3), ("U", 100), ("V", 101)):
                eng.set_tensor(tid, o.p[n])
        for ep in range(2):
            perm = rng.permutation(U)
            dl0, gl0 = plain.train_epoch(perm, 1, 1)
            dl1, gl1 = sh.train_epoch(perm, 1, 1)
            np.testing.assert_array_equal(dl0, dl1)
            np.testing.assert_allclose(gl0, gl1, rtol=1e-6)
        for tid in (0, 1, 2, 3, 100, 101):
            np.testing.assert_array_equal(plain.get_tensor(tid), sh.get_tensor(tid))
        np.testing.assert_array_equal(plain.scores(np.arange(10)), sh.scores(np.arange(10)))
    finally:
        sh.close()
        plain.close()


def test_two_rank_process_backend():
    """TWO rank processes on two real devices over an RCCL communicator of world size 2 -- the production form of the sharded fit
    (GANRec/GANMF.py:146-149 is the single session it replaces) -- against the union-batch fp64 oracle and the plain single-GPU
    engine: the owner-split schedule of ShardedEngine makes the union over ranks of step i the reference's minibatch i.  Skipped on
    a one-GPU box (RCCL takes one rank per device); the day a box with two devices runs the suite, this path is checked."""
    from ganmf_amd import _lib as L
    from ganmf_amd.dist import ShardedEngine
    from ganmf_amd.engine import Engine
    if L.load_library().ganmf_device_count() < 2:
        pytest.skip("needs two GPUs: one RCCL rank per device")
    rng = np.random.RandomState(12)
    U, N, k, e, B = 301, 420, 13, 40, 64
    urm = _urm(rng, U, N, 0.06)
    hp = dict(d_lr=1e-3, g_lr=1e-3, d_reg=1e-4, g_reg=0.0, m=5.0, recon_coefficient=0.1)
    o = GANMFOracle(U, N, k, e, dtype=np.float64, seed=4, **hp)
    w0 = o.get_params()
    sh = ShardedEngine(U, N, k, e, B, devices=[0, 1], backend="process", **hp)
    plain = Engine(U, N, k, e, B, **hp)
    try:
        assert sh.world == 2
        for eng in (sh, plain):
            eng.set_urm(urm)
            for n, tid in (("We", 0), ("be", 1), ("Wd", 2), ("bd", 3), ("U", 100), ("V", 101)):
                eng.set_tensor(tid, w0[n])
        for ep in range(3):
            perm = rng.permutation(U)
            dl_ref, gl_ref = o.train_epoch(urm, perm, B)
            dl0, gl0 = plain.train_epoch(perm, 1, 1)
            dl1, gl1 = sh.train_epoch(perm, 1, 1)
            np.testing.assert_allclose(dl1, dl_ref, rtol=1e-4)
            np.testing.assert_allclose(gl1, gl_ref, rtol=1e-4)
            np.testing.assert_allclose(dl1, dl0, rtol=1e-5)
        for n, tid in (("We", 0), ("be", 1), ("Wd", 2), ("bd", 3), ("U", 100), ("V", 101)):
            got = sh.get_tensor(tid)
            assert _err(got, o.p[n]) <= 1e-4, n
            assert _err(got, plain.get_tensor(tid).astype(np.float64)) <= 1e-4, n
        assert _err(sh.scores(np.arange(32)), o.scores(np.arange(32))) <= 1e-4
        # every rank holds the same replicated tensors, bit for bit (one writer per slice, then all-gather)
        for tid in (0, 1, 2, 3, 101):
            copies = sh._all("get_tensor", tid)
            np.testing.assert_array_equal(copies[0], copies[1])
    finally:
        sh.close()
        plain.close()
