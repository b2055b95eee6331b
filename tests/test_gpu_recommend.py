"""Device-side recommend (ganmf_set_seen_csr / ganmf_recommend) against the host route of
Base/BaseRecommender.py:155-247 (scores -> seen to -inf -> top-k) and the two evaluators on the same model."""
import json
import os

import numpy as np
import pytest
import scipy.sparse as sps

pytestmark = pytest.mark.gpu


def _model(mode, n_users, n_items, k, seed, density=0.08):
    from ganmf_amd.GANMF import GANMF
    rng = np.random.RandomState(seed)
    m = (rng.rand(n_users, n_items) < density).astype(np.float32)
    m[np.arange(n_users), rng.randint(0, n_items, n_users)] = 1.0
    urm = sps.csr_matrix(m)
    model = GANMF(urm, mode=mode, is_experiment=True)
    model._build(k, 16, 32)
    nu, ni = model.num_users, model.num_items
    model.engine.set_tensor(100, rng.randn(nu, k).astype(np.float32))
    model.engine.set_tensor(101, rng.randn(ni, k).astype(np.float32))
    model.URM_train = model._URM_eval
    return model, urm, rng


def _host_topk(model, users, cutoff, remove_seen):
    """the reference's own steps on the full score rows, ties to the smaller id"""
    scores = model._compute_item_score(users).astype(np.float32)
    if remove_seen:
        urm = model._URM_eval
        for i, u in enumerate(users):
            scores[i, urm.indices[urm.indptr[u]:urm.indptr[u + 1]]] = -np.inf
    order = np.lexsort((np.arange(scores.shape[1])[None, :].repeat(len(users), 0), -scores), axis=1)[:, :cutoff]
    vals = np.take_along_axis(scores, order, axis=1)
    return np.where(np.isfinite(vals), order, -1), vals


@pytest.mark.parametrize("mode", ["user", "item"])
@pytest.mark.parametrize("remove_seen", [True, False])
@pytest.mark.parametrize("cutoff", [1, 5, 50])
def test_device_topk_equals_host(mode, remove_seen, cutoff):
    model, urm, rng = _model(mode, 211, 333, 9, 3)
    users = rng.permutation(211)[:97]
    items, vals = model.engine.recommend(users, cutoff, transposed=(mode == "item"), remove_seen=remove_seen)
    ref_items, ref_vals = _host_topk(model, users, cutoff, remove_seen)
    assert np.array_equal(items, ref_items)
    assert np.array_equal(vals, ref_vals)          # same GEMM, same bits
    lists = model.recommend(users, cutoff=cutoff, remove_seen_flag=remove_seen)
    assert lists == [r[r >= 0].tolist() for r in ref_items]
    assert model.recommend(int(users[0]), cutoff=cutoff, remove_seen_flag=remove_seen) == lists[0]


def test_short_lists_padded_and_ties():
    from ganmf_amd.GANMF import GANMF
    n_users, n_items, k = 9, 70, 4
    m = np.zeros((n_users, n_items), np.float32)
    m[0, :] = 1.0                 # user 0 has seen everything
    m[1, 3:] = 1.0                # user 1: three items left
    m[2:, 0] = 1.0
    model = GANMF(sps.csr_matrix(m), mode="user", is_experiment=True)
    model._build(k, 8, 4)
    U = np.ones((n_users, k), np.float32)
    V = np.ones((n_items, k), np.float32)     # every score equal: ties resolve to increasing item id
    model.engine.set_tensor(100, U)
    model.engine.set_tensor(101, V)
    items, vals = model.engine.recommend(np.arange(n_users), 5, remove_seen=True)
    assert items[0].tolist() == [-1] * 5 and np.all(np.isneginf(vals[0]))
    assert items[1].tolist() == [0, 1, 2, -1, -1]
    assert items[2].tolist() == [1, 2, 3, 4, 5]
    model.URM_train = model._URM_eval
    lists = model.recommend(np.arange(n_users), cutoff=5)
    assert lists[0] == [] and lists[1] == [0, 1, 2] and lists[4] == [1, 2, 3, 4, 5]


def test_wide_rows_use_global_path():
    """score rows wider than the LDS staging limit (32768 floats) are selected in place"""
    model, urm, rng = _model("user", 6, 40000, 3, 11, density=0.001)
    users = np.array([5, 0, 3])
    items, vals = model.engine.recommend(users, 7, remove_seen=True)
    ref_items, ref_vals = _host_topk(model, users, 7, True)
    assert np.array_equal(items, ref_items) and np.array_equal(vals, ref_vals)


def test_recommend_errors():
    from ganmf_amd import _lib as L
    from ganmf_amd.engine import Engine
    eng = Engine(10, 20, 3, 4, 4)
    with pytest.raises(L.GanmfError, match="remove_seen needs"):
        eng.recommend(np.arange(3), 5, remove_seen=True)
    items, _ = eng.recommend(np.arange(3), 5, remove_seen=False)      # allowed without a seen matrix
    assert items.shape == (3, 5)
    with pytest.raises(L.GanmfError, match="cutoff"):
        eng.recommend(np.arange(3), 21, remove_seen=False)
    wide = Engine(4, 3000, 2, 4, 4)
    with pytest.raises(L.GanmfError, match="cutoff"):     # GANMF_RECOMMEND_MAX_CUTOFF
        wide.recommend(np.arange(2), 1025, remove_seen=False)
    assert wide.recommend(np.arange(2), 1024, remove_seen=False)[0].shape == (2, 1024)
    wide.close()
    with pytest.raises(L.GanmfError, match="out of range"):
        eng.recommend(np.array([10]), 5, remove_seen=False)
    bad = sps.csr_matrix(np.ones((10, 21), np.float32))
    eng.set_seen(bad)
    with pytest.raises(L.GanmfError, match="remove_seen needs"):
        eng.recommend(np.arange(3), 5, remove_seen=True)
    eng.close()


def test_fast_evaluator_on_device_matches_slow(golden_dir):
    """Whole-protocol check on ML-1M shapes: the device top-k + block evaluator returns the metrics of the
    reference-order evaluator (which pulls every score row to the host)."""
    from ganmf_amd.GANMF import GANMF
    from ganmf_amd.evaluation import EvaluatorHoldout, EvaluatorHoldoutFast
    train = sps.load_npz(os.path.join(golden_dir, "Movielens1M_URM_train.npz")).tocsr()
    test = sps.load_npz(os.path.join(golden_dir, "Movielens1M_URM_test.npz")).tocsr()
    for mode in ("user", "item"):
        model = GANMF(train, mode=mode, is_experiment=True, seed=3)
        model.fit(num_factors=16, emb_dim=32, epochs=2, batch_size=128, d_lr=1e-3, g_lr=1e-3, m=10, recon_coefficient=0.1)
        slow, _ = EvaluatorHoldout(test, [5, 20]).evaluateRecommender(model)
        fast, _ = EvaluatorHoldoutFast(test, [5, 20]).evaluateRecommender(model)
        for c in (5, 20):
            for k, v in slow[c].items():
                if k == "RMSE":
                    assert np.isnan(fast[c][k])
                    continue
                assert abs(fast[c][k] - v) <= 1e-9 + 2e-5 * abs(v), (mode, c, k, fast[c][k], v)


def test_save_load_model_round_trip_in_reference_format(tmp_path, golden_dir):
    """saveModel writes build_params.pkl + the Saver bundle (GANMF.py:309-314); loadModel (GANMF.py:316-339)
    restores identical scores; the index carries the reference's tensor names."""
    import pickle
    from ganmf_amd import tf_bundle
    from ganmf_amd.DisGANMF import DisGANMF
    model, urm, rng = _model("item", 57, 91, 6, 21)
    users = np.arange(57)
    before = model._compute_item_score(users)
    model.saveModel(str(tmp_path))
    assert pickle.load(open(tmp_path / "build_params.pkl", "rb")) == {"num_factors": 6, "emb_dim": 16}
    _, entries = tf_bundle.read_index(str(tmp_path / "GANMF_item.index"))
    _, ref_entries = tf_bundle.read_index(os.path.join(golden_dir, "kat1_GANMF_item.index"))
    assert list(entries) == list(ref_entries)
    from ganmf_amd.GANMF import GANMF
    again = GANMF(urm, mode="item", is_experiment=True)
    again.loadModel(str(tmp_path))
    assert np.array_equal(again._compute_item_score(users), before)
    assert again.recommend(users, cutoff=5) == model.recommend(users, cutoff=5)
    with pytest.raises(FileNotFoundError):
        again.loadModel(str(tmp_path), file_name="missing")

    d = DisGANMF(urm, mode="user", is_experiment=True, seed=2)
    d.fit(num_factors=4, d_layers=2, d_nodes=8, d_hidden_act="tanh", epochs=1, batch_size=16)
    d.saveModel(str(tmp_path), "dis")
    keys = list(tf_bundle.read_index(str(tmp_path / "dis.index"))[1])
    assert keys == ["discriminator/D_output/bias", "discriminator/D_output/kernel", "discriminator/layer_0/bias",
                    "discriminator/layer_0/kernel", "discriminator/layer_1/bias", "discriminator/layer_1/kernel",
                    "generator/item_embeddings", "generator/user_embeddings"]
    d2 = DisGANMF(urm, mode="user", is_experiment=True)
    d2.load_bundle(str(tmp_path), "dis", 4, 2, 8, "tanh")
    assert np.array_equal(d2._compute_item_score(users[:9]), d._compute_item_score(users[:9]))


def test_trial_parallel_tuner_on_gpu(tmp_path, golden_dir):
    """ganmf_amd.tune with the real GANMF class: two worker processes share the GPU, each running whole trials
    (fit with early stopping on the device evaluator, validation, engine teardown)."""
    from GANRec.GANMF import GANMF
    from ganmf_amd import tune
    train = sps.load_npz(os.path.join(golden_dir, "hetrec2011_URM_train_small.npz")).tocsr()[:600, :900]
    val = sps.load_npz(os.path.join(golden_dir, "hetrec2011_URM_validation.npz")).tocsr()[:600, :900]
    t = tune.TrialParallelTuner(GANMF, train, val, val, str(tmp_path / "tune"), seed=3, n_workers=2, devices=[0])
    for d in t.dims:                       # keep the trials short
        if d.name == "epochs":
            d.choices = [12]
        if d.name in ("num_factors", "emb_dim"):
            d.high = 32
    best, params = t.tune(evals=4, verbose=False)
    assert len(t.func_vals) == 4 and all(np.isfinite(t.func_vals))
    assert "trial failed" not in open(os.path.join(str(tmp_path / "tune"), "results.txt")).read()
    assert best < 0 and set(params) == {d.name for d in t.dims}


@pytest.mark.parametrize("mode", ["user", "item"])
@pytest.mark.parametrize("exclude_seen", [True, False])
def test_device_metrics_equal_host_metrics(mode, exclude_seen):
    """ganmf_evaluate (metric sums formed on the device from the device's own top-k lists) against the same
    evaluator computing them on the host from the downloaded ids: graded ratings (DCG gains), users whose lists are shorter
    than the cut-off (few unseen items left), users without test items dropped by the evaluator, four cut-offs at once."""
    from ganmf_amd.evaluation import EvaluatorHoldoutFast
    model, urm, rng = _model(mode, 300, 23, 6, seed=11, density=0.5)      # 23 items, half of them seen: lists of ~11 < cutoff 20
    nu, ni = urm.shape
    t = (rng.rand(nu, ni) < 0.15) * rng.randint(1, 6, size=(nu, ni))
    t[rng.rand(nu) < 0.1] = 0                                             # users without test items
    test = sps.csr_matrix(t.astype(np.float32))
    ev = EvaluatorHoldoutFast(test, [1, 5, 10, 20], exclude_seen=exclude_seen)
    assert ev.use_device_metrics
    dev, _ = ev.evaluateRecommender(model)
    ev.use_device_metrics = False
    host, _ = ev.evaluateRecommender(model)
    for c in (1, 5, 10, 20):
        for name, v in host[c].items():
            if name == "RMSE":
                assert np.isnan(dev[c][name])
                continue
            assert abs(dev[c][name] - v) <= 1e-12 * max(1.0, abs(v)), (mode, exclude_seen, c, name, dev[c][name], v)
    assert host[5]["MAP"] > 0 and host[20]["NDCG"] > 0
    model.engine.close()


def test_device_metrics_reject_bad_input():
    from ganmf_amd._lib import GanmfError
    model, urm, rng = _model("user", 40, 30, 4, seed=3)
    eng = model.engine
    test = sps.csr_matrix((rng.rand(40, 30) < 0.2).astype(np.float32))
    test.sort_indices()
    disc, ideal = np.ones(5), np.ones((4, 5))
    with pytest.raises(GanmfError):       # no test matrix yet
        eng.evaluate(np.arange(4), [5], disc, ideal)
    eng.set_test(test, np.ones(test.nnz))
    with pytest.raises(GanmfError):       # more cut-offs than one call takes
        eng.evaluate(np.arange(4), list(range(1, 10)), np.ones(9), np.ones((4, 9)))
    bad = test.copy()
    bad.indices = bad.indices[::-1].copy()
    with pytest.raises((GanmfError, AssertionError)):
        eng.set_test(bad, np.ones(bad.nnz))
    out = eng.evaluate(np.arange(4), [5], disc, ideal, remove_seen=False)
    assert out.shape == (1, 9) and np.all(np.isfinite(out))
    eng.close()


@pytest.mark.parametrize("mode", ["user", "item"])
def test_mf_contract_items_to_compute_and_cold_users(mode, golden_dir):
    """BOTH scoring contracts through the HIP path on one matrix with cold training rows (fixture: oracle/make_golden.py::
    mf_contract_golden, expected outputs produced by the reference's own classes):
      score_contract="mf"    Base/BaseMatrixFactorizationRecommender.py:113-119,128-143 -- `items_to_compute` masks the other
                             items, users without a training interaction score -inf everywhere and are recommended nothing;
      score_contract="ganmf" (default) GANRec/GANMF.py:285-292 -- plain U[ids] . V^T for every user, `items_to_compute` ignored:
                             a cold training row is recommended items.
    Scores, the device top-k route, the host route and the device evaluator.  Item mode scores the same evaluation matrix from
    the transposed model (generator rows = items)."""
    from ganmf_amd.GANMF import GANMF
    from ganmf_amd.evaluation import EvaluatorHoldoutFast
    g = np.load(os.path.join(golden_dir, "mf_contract.npz"))
    n_users, n_items = (int(x) for x in g["urm_shape"])
    urm = sps.csr_matrix((np.ones(len(g["urm_indices"]), np.float32), g["urm_indices"], g["urm_indptr"]), shape=(n_users, n_items))
    k = g["U"].shape[1]
    users, items, cold = g["users"], g["items_to_compute"], g["cold_users"]

    def build(**kw):
        model = GANMF(urm, mode=mode, is_experiment=True, **kw)
        model._build(k, 16, 32)
        # the evaluation-orientation factors of the fixture: in item mode the generator's "users" are the catalogue items
        model.engine.set_tensor(100, g["U"] if mode == "user" else g["V"])
        model.engine.set_tensor(101, g["V"] if mode == "user" else g["U"])
        model.URM_train = model._URM_eval
        return model

    def same(got, want):
        assert np.array_equal(np.isneginf(got), np.isneginf(want))
        fin = np.isfinite(want)
        assert np.abs(got[fin] - want[fin]).max() <= 2e-6 * np.abs(want[fin]).max()

    pad = lambda lists: np.array([l + [-1] * (10 - len(l)) for l in lists], dtype=np.int32)

    # ---- the MF contract ------------------------------------------------------------------------------------------------------
    model = build(score_contract="mf")
    same(model._compute_item_score(users), g["scores_all"])                                   # cold users: -inf everywhere
    same(model._compute_item_score(users, items_to_compute=items), g["scores_subset"])        # other items: -inf
    assert np.all(np.isneginf(model._compute_item_score(cold)))
    # device top-k route (scores never leave the GPU) and the host route (return_scores) agree with the reference's lists
    assert np.array_equal(pad(model.recommend(users, cutoff=10, remove_seen_flag=True)), g["rank_all_seen"])
    assert np.array_equal(pad(model.recommend(users, cutoff=10, remove_seen_flag=True, items_to_compute=items)), g["rank_subset_seen"])
    assert np.array_equal(pad(model.recommend(users, cutoff=10, remove_seen_flag=False, items_to_compute=items)), g["rank_subset_unseen"])
    host_lists, _ = model.recommend(users, cutoff=10, remove_seen_flag=True, items_to_compute=items, return_scores=True)
    assert np.array_equal(pad(host_lists), g["rank_subset_seen"])
    assert all(len(l) == 0 for l in model.recommend(cold, cutoff=10))
    # the filter does not leak into later calls
    same(model._compute_item_score(users), g["scores_all"])
    # engine-level errors: an item beyond the score width, cold masking without the seen matrix
    from ganmf_amd import _lib as L
    with pytest.raises(L.GanmfError):
        model.engine.set_score_filter([10 ** 6], mask_cold=True)
    mf = model

    # ---- the reference GANMF's own contract (the default) ---------------------------------------------------------------------
    model = build()
    assert model.score_contract == "ganmf"
    want = g["ganmf_scores_all"]
    assert np.all(np.isfinite(want))
    same(model._compute_item_score(users), want)
    same(model._compute_item_score(users, items_to_compute=items), want)                      # accepted and ignored
    assert np.all(np.isfinite(model._compute_item_score(cold)))
    assert np.array_equal(pad(model.recommend(users, cutoff=10, remove_seen_flag=True)), g["ganmf_rank_all_seen"])
    assert np.array_equal(pad(model.recommend(users, cutoff=10, remove_seen_flag=True, items_to_compute=items)), g["ganmf_rank_subset_seen"])
    assert np.array_equal(pad(model.recommend(users, cutoff=10, remove_seen_flag=False)), g["ganmf_rank_all_unseen"])
    host_lists, _ = model.recommend(users, cutoff=10, remove_seen_flag=True, items_to_compute=items, return_scores=True)
    assert np.array_equal(pad(host_lists), g["ganmf_rank_all_seen"])
    assert all(len(l) == 10 for l in model.recommend(cold, cutoff=10))                        # a cold training row IS recommended items

    # ---- device evaluator: test items of the cold rows count under "ganmf" and cannot be hit under "mf" -------------------------
    # test matrix: for every fixture user, the first three items of the reference GANMF's own list -> each of them is a hit
    t = sps.lil_matrix((n_users, n_items), dtype=np.float32)
    for u, row in zip(users, g["ganmf_rank_all_seen"]):
        t[int(u), [int(i) for i in row[:3]]] = 1.0
    test = t.tocsr()
    res = {}
    for name, m in (("ganmf", model), ("mf", mf)):
        ev = EvaluatorHoldoutFast(test, [5])
        assert ev.use_device_metrics
        dev, _ = ev.evaluateRecommender(m)
        ev.use_device_metrics = False
        host, _ = ev.evaluateRecommender(m)
        for key in ("PRECISION", "RECALL", "MAP", "NDCG"):
            assert abs(dev[5][key] - host[5][key]) <= 1e-12, (name, key)
        res[name] = dev[5]
    n_cold = sum(1 for u in users if u in set(cold.tolist()))
    assert n_cold == 3
    assert abs(res["ganmf"]["RECALL"] - 1.0) <= 1e-12                                          # every user's three test items are in the top five
    assert abs(res["mf"]["RECALL"] - (len(users) - n_cold) / len(users)) <= 1e-12             # the cold rows are recommended nothing
    # environment switch (the reference's drivers construct the class themselves)
    os.environ["GANMF_SCORE_CONTRACT"] = "mf"
    try:
        assert GANMF(urm, mode=mode, is_experiment=True).score_contract == "mf"
    finally:
        del os.environ["GANMF_SCORE_CONTRACT"]
    with pytest.raises(ValueError):
        GANMF(urm, mode=mode, is_experiment=True, score_contract="other")
    model.engine.close()
    mf.engine.close()


def test_concurrent_engines_bit_identical(golden_dir):
    """Two fits in two threads of one process (two engines, two HIP streams: the tuner's engines_per_worker) leave exactly the
    tensors and losses each leaves when it runs alone -- GANMF and DisGANMF side by side."""
    import threading
    from GANRec.DisGANMF import DisGANMF
    from GANRec.GANMF import GANMF
    urm = sps.load_npz(os.path.join(golden_dir, "tiny_urm.npz")).tocsr()

    def fit(kind, out, key):
        if kind == "ganmf":
            m = GANMF(urm, mode="user", seed=3, is_experiment=True)
            m.schedule_rng = np.random.RandomState(11)
            m.fit(num_factors=9, emb_dim=24, epochs=6, batch_size=16, d_lr=1e-3, g_lr=1e-3, d_reg=1e-4, m=3, recon_coefficient=0.05)
        else:
            m = DisGANMF(urm, mode="item", seed=4, is_experiment=True)
            m.schedule_rng = np.random.RandomState(12)
            m.fit(num_factors=7, d_nodes=20, d_layers=2, d_hidden_act="tanh", epochs=6, batch_size=16, d_lr=1e-3, g_lr=1e-3)
        out[key] = (m.user_factors().copy(), m.item_factors().copy(), list(m.train_d_loss), list(m.train_g_loss))
        m.engine.close()

    solo, both = {}, {}
    for kind in ("ganmf", "dis"):
        fit(kind, solo, kind)
    for _ in range(3):      # several rounds: an interleaving-dependent result would not survive them
        threads = [threading.Thread(target=fit, args=(kind, both, kind)) for kind in ("ganmf", "dis")]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        for kind in ("ganmf", "dis"):
            for a, b in zip(solo[kind][:2], both[kind][:2]):
                assert np.array_equal(a, b), kind
            assert solo[kind][2] == both[kind][2] and solo[kind][3] == both[kind][3], kind
