"""Why the end-to-end DisGANMF comparison (tests/test_gpu_statistical.py) uses distribution-wide bands: (1) two builds of the SAME
update that differ by rounding only (output-layer Adam inside / outside the backward-top kernel: bit-identical after one D and one G
step) separate exponentially over the epochs; (2) the factor norms of full ML-1M trainings spread over initialisations.
usage: python tools/chaos_check.py [norms]"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if "norms" in sys.argv:
    import scipy.sparse as sps
    from ganmf_amd.DisGANMF import DisGANMF
    from ganmf_amd.evaluation import EvaluatorHoldoutFast
    g = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
    kat = json.load(open(os.path.join(g, "statistical_kat_disganmf_ml1m_user.json")))
    train = sps.load_npz(os.path.join(g, "Movielens1M_URM_train.npz")).tocsr()
    test = sps.load_npz(os.path.join(g, "Movielens1M_URM_test.npz")).tocsr()
    ev = EvaluatorHoldoutFast(test, [5])
    for seed in (1337, 1, 2, 3, 4, 5, 6, 7):
        np.random.seed(seed)
        m = DisGANMF(train, mode="user", seed=seed, is_experiment=True)
        m.fit(validation_set=None, sample_every=None, validation_evaluator=None, **kat["best_params"])
        r = ev.evaluateRecommender(m)[0][5]
        print("seed %4d  MAP@5 %.4f NDCG@5 %.4f  |U| %.3f |V| %.3f" % (seed, r["MAP"], r["NDCG"], np.linalg.norm(m.user_factors()),
                                                                        np.linalg.norm(m.item_factors())), flush=True)
        m.engine.close()
    sys.exit(0)

from ganmf_amd import _lib as L  # noqa: E402
from ganmf_amd.engine import Engine  # noqa: E402
from ganmf_amd.synthetic import synthetic_urm  # noqa: E402
from oracle.ganmf_oracle import DisGANMFOracle  # noqa: E402   (initial weights only)

U, N, k, e, B = 6040, 3706, 117, 1024, 128
hp = dict(d_lr=1e-4, g_lr=5.665e-4, d_reg=3.002e-5, g_reg=0.0, recon_coefficient=0.5)
urm = synthetic_urm(U, N, 0.035, seed=1337)
o = DisGANMFOracle(U, N, k, d_layers=1, d_nodes=e, d_hidden_act="linear", dtype=np.float32, seed=1337, **hp)
IDS = {"U": 100, "V": 101, "Wo": 2, "bo": 3, "W0": 0, "b0": 1}


def make(fuse):
    os.environ["GANMF_MULTI"] = str(63 + 64 * int(fuse))      # bit 6: hidden layers' Adam fused
    eng = Engine(U, N, k, e, B, model=L.MODEL_DISGANMF, d_layers=1, d_act="linear", m=0.0, **hp)
    eng.set_urm(urm)
    for n, tid in IDS.items():
        eng.set_tensor(tid, o.p[n])
    return eng


a, b = make("1"), make("0")
rng = np.random.RandomState(0)


def diff(tag):
    out = []
    for n, tid in IDS.items():
        x, y = a.get_tensor(tid).astype(np.float64), b.get_tensor(tid).astype(np.float64)
        out.append("%s %.2e" % (n, np.max(np.abs(x - y)) / (np.max(np.abs(y)) + 1e-30)))
    print(tag, " ".join(out), "| |U| %.4f %.4f" % (np.linalg.norm(a.get_tensor(100)), np.linalg.norm(b.get_tensor(100))), flush=True)


perm = rng.permutation(U)
a.train_step(0, perm[:B]); b.train_step(0, perm[:B]); diff("1 D step   ")
a.train_step(1, perm[:B]); b.train_step(1, perm[:B]); diff("+1 G step  ")
for ep in range(1, 41):
    perm = rng.permutation(U)
    la = a.train_epoch(perm, 1, 1); lb = b.train_epoch(perm, 1, 1)
    if ep in (1, 2, 5, 10, 20, 40):
        diff("epoch %2d   " % ep)
