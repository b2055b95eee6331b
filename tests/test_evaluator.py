"""The build's evaluator / recommend() against golden outputs of the reference's own
Base/Evaluation/Evaluator.py + BaseRecommender.recommend (oracle/make_golden.py), CPU only."""
import json
import os

import numpy as np
import scipy.sparse as sps

from ganmf_amd.base import BaseRecommender
from ganmf_amd.evaluation import EvaluatorHoldout


class _Factors(BaseRecommender):
    def __init__(self, urm, U, V):
        super().__init__(urm)
        self.U, self.V = U, V

    def _compute_item_score(self, user_id_array, items_to_compute=None):
        return self.U[user_id_array] @ self.V.T


def test_evaluator_matches_reference_golden(golden_dir):
    f = np.load(os.path.join(golden_dir, "evaluator_factors.npz"))
    exp = json.load(open(os.path.join(golden_dir, "evaluator_expected.json")))
    train = sps.load_npz(os.path.join(golden_dir, "hetrec2011_URM_train_small.npz")).tocsr()
    val = sps.load_npz(os.path.join(golden_dir, "hetrec2011_URM_validation.npz")).tocsr()
    res, text = EvaluatorHoldout(val, [5, 10]).evaluateRecommender(_Factors(train, f["U"], f["V"]))
    assert "CUTOFF: 5" in text
    for c, d in exp.items():
        for k in ("ROC_AUC", "PRECISION", "PRECISION_RECALL_MIN_DEN", "RECALL", "MAP", "MRR", "NDCG", "F1", "HIT_RATE",
                  "ARHR", "RMSE"):
            assert abs(res[int(c)][k] - d[k]) <= 1e-9 + 2e-6 * abs(d[k]), (c, k, res[int(c)][k], d[k])


def test_fast_evaluator_matches_reference_golden(golden_dir):
    """EvaluatorHoldoutFast (block-vectorised, top-k ids only) against the same reference golden; RMSE is
    not available without the full score rows and must be NaN.  The golden itself carries the reference's
    float32 running sums over ~2100 users (a few 1e-6 relative); the fast path sums in float64, hence 2e-5."""
    from ganmf_amd.evaluation import EvaluatorHoldoutFast
    f = np.load(os.path.join(golden_dir, "evaluator_factors.npz"))
    exp = json.load(open(os.path.join(golden_dir, "evaluator_expected.json")))
    train = sps.load_npz(os.path.join(golden_dir, "hetrec2011_URM_train_small.npz")).tocsr()
    val = sps.load_npz(os.path.join(golden_dir, "hetrec2011_URM_validation.npz")).tocsr()
    res, text = EvaluatorHoldoutFast(val, [5, 10]).evaluateRecommender(_Factors(train, f["U"], f["V"]))
    assert "CUTOFF: 10" in text
    for c, d in exp.items():
        for k in ("ROC_AUC", "PRECISION", "PRECISION_RECALL_MIN_DEN", "RECALL", "MAP", "MRR", "NDCG", "F1", "HIT_RATE",
                  "ARHR"):
            assert abs(res[int(c)][k] - d[k]) <= 1e-9 + 2e-5 * abs(d[k]), (c, k, res[int(c)][k], d[k])
        assert np.isnan(res[int(c)]["RMSE"])


def test_fast_evaluator_short_lists_and_ratings():
    """Users whose recommendation list is shorter than the cutoff (everything else seen) and graded test
    ratings: the two evaluators must agree metric by metric."""
    from ganmf_amd.evaluation import EvaluatorHoldoutFast
    rng = np.random.RandomState(5)
    n_users, n_items = 40, 12
    train = (rng.rand(n_users, n_items) < 0.5).astype(np.float32)
    train[:6, :] = 1.0
    train[:6, :3] = 0.0          # 3 unseen items < cutoff 5
    train[6, :] = 1.0            # nothing left to recommend
    test = ((rng.rand(n_users, n_items) < 0.3) & (train == 0)) * rng.randint(1, 6, size=(n_users, n_items))
    test[6, 0] = 3               # relevant item that can never be recommended
    U, V = rng.randn(n_users, 4).astype(np.float32), rng.randn(n_items, 4).astype(np.float32)
    rec = _Factors(sps.csr_matrix(train), U, V)
    test = sps.csr_matrix(test.astype(np.float32))
    slow, _ = EvaluatorHoldout(test, [1, 5, 8]).evaluateRecommender(rec)
    fast, _ = EvaluatorHoldoutFast(test, [1, 5, 8]).evaluateRecommender(rec)
    for c in (1, 5, 8):
        for k, v in slow[c].items():
            if k == "RMSE":
                continue
            assert abs(fast[c][k] - v) <= 1e-9 + 2e-6 * abs(v), (c, k, fast[c][k], v)


def test_kat1_ranking_on_cpu(golden_dir):
    """KAT-1 without a GPU: checkpoint tensors scored in numpy through the build's recommend()."""
    t = np.load(os.path.join(golden_dir, "kat1_checkpoint_tensors.npz"))
    exp = json.load(open(os.path.join(golden_dir, "kat1_expected.json")))
    train = sps.load_npz(os.path.join(golden_dir, "LastFM_URM_train.npz")).tocsr()
    # item mode: evaluation users are the generator's items
    rec = _Factors(train, t["V"], t["U"])
    users = np.array(exp["users"])
    ranking = rec.recommend(users, cutoff=50, remove_seen_flag=True)
    assert ranking == exp["ranking_top50"]
    single = rec.recommend(int(users[3]), cutoff=50)
    assert single == exp["ranking_top50"][3]


def test_early_stopping_protocol():
    from ganmf_amd.early_stopping import EarlyStoppingScheduler

    class M:
        def __init__(self):
            self.saved = self.loaded = self.stopped = 0

        def save_current_model(self):
            self.saved += 1

        def load_model(self):
            self.loaded += 1

        def stop_fit(self):
            self.stopped += 1

    class E:
        def __init__(self, seq):
            self.seq = list(seq)

        def evaluateRecommender(self, model):
            return {5: {"MAP": self.seq.pop(0)}}, ""

    m = M()
    es = EarlyStoppingScheduler(m, E([0.1, 0.2, 0.15, 0.2, 0.19]), metrics=["MAP"], freq=1, allow_worse=2, after=0)
    for ep in range(1, 6):
        es(ep)
    # improvements at epochs 1, 2; worse (<=) at 3, 4 -> counter 2 -> 0; epoch 5 worse with none left -> stop + restore
    assert (m.saved, m.stopped, m.loaded) == (2, 1, 1)


def test_all_miss_recommender_scores_zero_f1():
    """A recommender that never hits: every accuracy metric is 0 and the F1 key exists (the reference's empty
    metrics dict keeps F1 = 0.0), so early stopping on 'F1' does not raise."""
    from ganmf_amd.evaluation import EvaluatorHoldoutFast
    n_users, n_items = 9, 20
    test = sps.csr_matrix((np.ones(n_users, dtype=np.float32), (np.arange(n_users), np.zeros(n_users, dtype=int))),
                          shape=(n_users, n_items))
    train = sps.csr_matrix((n_users, n_items), dtype=np.float32)
    # item 0 (the only test item) always gets the lowest score -> never inside the top 5
    rec = _Factors(train, np.ones((n_users, 1), dtype=np.float32), np.arange(n_items, dtype=np.float32)[:, None])
    for cls in (EvaluatorHoldout, EvaluatorHoldoutFast):
        res, _ = cls(test, [5]).evaluateRecommender(rec)
        assert res[5]["F1"] == 0.0 and res[5]["MAP"] == 0.0 and res[5]["PRECISION"] == 0.0, (cls, res)
