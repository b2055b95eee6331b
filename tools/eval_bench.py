"""Hold-out evaluation wall time on ML-1M shapes: reference-order evaluator (all score rows to the host, per-user
python metrics) vs device top-k + block evaluator.  Usage: python tools/eval_bench.py"""
import os
import sys
import time

import numpy as np
import scipy.sparse as sps

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ganmf_amd.GANMF import GANMF  # noqa: E402
from ganmf_amd.evaluation import EvaluatorHoldout, EvaluatorHoldoutFast  # noqa: E402

g = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
train = sps.load_npz(os.path.join(g, "Movielens1M_URM_train.npz")).tocsr()
test = sps.load_npz(os.path.join(g, "Movielens1M_URM_test.npz")).tocsr()
model = GANMF(train, mode="user", is_experiment=True, seed=1)
model.fit(num_factors=250, emb_dim=992, epochs=1, batch_size=128, d_lr=1e-3, g_lr=1e-3, m=10, recon_coefficient=0.1)
slow_ev, fast_ev, host_ev = EvaluatorHoldout(test, [5]), EvaluatorHoldoutFast(test, [5]), EvaluatorHoldoutFast(test, [5])
host_ev.use_device_metrics = False      # device top-k ids, metrics in numpy on the host (round 1's fast evaluator)
users = np.arange(train.shape[0])
for name, fn in (("slow evaluator", lambda: slow_ev.evaluateRecommender(model)),
                 ("fast evaluator, metrics on the host", lambda: host_ev.evaluateRecommender(model)),
                 ("fast evaluator, metrics on the device", lambda: fast_ev.evaluateRecommender(model)),
                 ("device recommend top-5, all users", lambda: model.recommend_topk(users, 5)),
                 ("device recommend top-50, all users", lambda: model.recommend_topk(users, 50)),
                 ("host recommend top-5, all users", lambda: model.recommend(users, cutoff=5, return_scores=True))):
    fn()
    t0 = time.time()
    for _ in range(3):
        fn()
    print("%-40s %8.1f ms" % (name, (time.time() - t0) / 3 * 1e3))
