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
