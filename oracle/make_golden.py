"""Generates tests/golden/* .  Runs ONLY in the build container (needs /root/reference).

Nothing here ships to the GPU box as code the tests execute; the tests read the emitted data.
The reference's own pure-numpy modules (Base/BaseRecommender.py, Base/Evaluation/*) are imported
from where they lie to produce expected outputs; no reference source is copied.

    python oracle/make_golden.py            # all fixtures
"""
import json
import os
import pickle
import shutil
import sys

import numpy as np
import scipy.sparse as sps
import numpy.ma  # noqa: F401  (must be imported before the alias shim)

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)


def _import_reference():
    # numpy>=1.24 dropped the aliases the reference still uses (BaseRecommender.py:31, Evaluator.py:155)
    np.int = int
    np.bool = np.bool_
    np.float = float
    sys.path.insert(0, REF)
    from Base.BaseRecommender import BaseRecommender
    from Base.Evaluation.Evaluator import EvaluatorHoldout
    return BaseRecommender, EvaluatorHoldout


def _clean(results):
    return {str(c): {k: float(v) for k, v in d.items()} for c, d in results.items()}


def kat1(BaseRecommender, EvaluatorHoldout):
    """Surviving TF V2-bundle checkpoint -> item-mode scores -> reference evaluator
    == stored test_results.pkl  (SURVEY F4, Appendix C)."""
    d = os.path.join(REF, "feature_matching/GANMF_item_LastFM_00/GANMF_item_LastFM")
    raw = np.fromfile(os.path.join(d, "GANMF_item.data-00000-of-00001"), dtype="<f4")
    assert raw.nbytes == 2090708
    # key-sorted tensor order, offsets in bytes (Appendix C)
    off = {"bd": (0, (1884,)), "Wd": (7536, (133, 1884)), "be": (1009824, (133,)),
           "We": (1010356, (1884, 133)), "V": (2012644, (1884, 1)), "U": (2020180, (17632, 1))}
    t = {k: raw[o // 4: o // 4 + int(np.prod(s))].reshape(s).copy() for k, (o, s) in off.items()}
    # only the generator tensors are needed for scoring; the autoencoder tensors stay behind
    np.savez_compressed(os.path.join(OUT, "kat1_checkpoint_tensors.npz"), U=t["U"], V=t["V"],
                        be=t["be"], bd=t["bd"])
    for split in ("train", "test"):
        shutil.copyfile(os.path.join(REF, "experiments/datasets/LastFM_URM_%s.npz" % split),
                        os.path.join(OUT, "LastFM_URM_%s.npz" % split))
    expected = pickle.load(open(os.path.join(d, "test_results.pkl"), "rb"))

    urm_train = sps.load_npz(os.path.join(OUT, "LastFM_URM_train.npz")).tocsr()
    urm_test = sps.load_npz(os.path.join(OUT, "LastFM_URM_test.npz")).tocsr()

    class CheckpointScorer(BaseRecommender):
        """item mode (GANMF.py:288-290): generator 'users' are the catalogue items."""
        RECOMMENDER_NAME = "kat1"

        def _compute_item_score(self, user_id_array, items_to_compute=None):
            return (t["U"] @ t["V"].T).T[user_id_array]

    rec = CheckpointScorer(urm_train)
    got, _ = EvaluatorHoldout(urm_test, [5, 10, 20, 50]).evaluateRecommender(rec)
    for c in expected:
        for k in expected[c]:
            # float32-accumulated metrics differ in the last ulp across numpy versions
            assert abs(got[c][k] - expected[c][k]) <= 1e-9 + 2e-7 * abs(expected[c][k]), (c, k, got[c][k], expected[c][k])
    users = np.arange(0, 1884, 97)
    ranking, scores = rec.recommend(users, cutoff=50, remove_seen_flag=True, return_scores=True)
    json.dump({"expected_metrics": _clean(expected), "users": users.tolist(), "ranking_top50": ranking},
              open(os.path.join(OUT, "kat1_expected.json"), "w"))
    print("KAT-1 reproduced: MAP@5 =", got[5]["MAP"], "RMSE =", got[5]["RMSE"])


def bundle_golden():
    """Checkpoint-format pin: the surviving TF V2 bundle is read by the
    build's reader, re-written by the build's writer and must come out byte for byte; the 354-byte .index
    (data, not source) and one data-less .index of another run are committed as fixtures."""
    import hashlib
    import tempfile
    from ganmf_amd import tf_bundle
    d = os.path.join(REF, "feature_matching/GANMF_item_LastFM_00/GANMF_item_LastFM/GANMF_item")
    tensors = tf_bundle.read_bundle(d)            # verifies every block and tensor CRC-32C
    tmp = tempfile.mkdtemp()
    tf_bundle.write_bundle(os.path.join(tmp, "GANMF_item"), tensors)
    sha = {}
    for ext in (".index", ".data-00000-of-00001"):
        ref, got = open(d + ext, "rb").read(), open(os.path.join(tmp, "GANMF_item" + ext), "rb").read()
        assert ref == got, "bundle writer does not reproduce the reference " + ext
        sha[ext] = hashlib.sha256(ref).hexdigest()
    shutil.copyfile(d + ".index", os.path.join(OUT, "kat1_GANMF_item.index"))
    shutil.copyfile(os.path.join(REF, "test_results/GANMF_user_1M/GANMF_user.index"),
                    os.path.join(OUT, "ml1m_GANMF_user.index"))
    _, e1m = tf_bundle.read_index(os.path.join(OUT, "ml1m_GANMF_user.index"))
    json.dump({"writer_reproduces_reference_bytes": True, "sha256": sha,
               "ml1m_user_entries": {k: {"shape": list(v["shape"]), "offset": v["offset"], "size": v["size"]}
                                     for k, v in e1m.items()}},
              open(os.path.join(OUT, "kat1_bundle_check.json"), "w"), indent=1)
    print("bundle: writer reproduces the reference .index/.data byte for byte")


def evaluator_golden(BaseRecommender, EvaluatorHoldout):
    """Non-degenerate regime for the build's own evaluator: a seeded random rank-8 model scored
    by the reference evaluator on the hetrec2011 validation split."""
    for split in ("train_small", "validation"):
        shutil.copyfile(os.path.join(REF, "experiments/datasets/Movielenshetrec2011_URM_%s.npz" % split),
                        os.path.join(OUT, "hetrec2011_URM_%s.npz" % split))
    urm_train = sps.load_npz(os.path.join(OUT, "hetrec2011_URM_train_small.npz")).tocsr()
    urm_val = sps.load_npz(os.path.join(OUT, "hetrec2011_URM_validation.npz")).tocsr()
    rng = np.random.RandomState(77)
    # popularity-biased so that metrics are far from zero
    pop = np.asarray(urm_train.sum(axis=0)).ravel().astype(np.float32)
    Uf = rng.rand(urm_train.shape[0], 8).astype(np.float32)
    Vf = (rng.rand(urm_train.shape[1], 8) * (1 + np.log1p(pop))[:, None]).astype(np.float32)

    class R(BaseRecommender):
        RECOMMENDER_NAME = "evalgold"

        def _compute_item_score(self, user_id_array, items_to_compute=None):
            return Uf[user_id_array] @ Vf.T

    got, _ = EvaluatorHoldout(urm_val, [5, 10]).evaluateRecommender(R(urm_train))
    np.savez_compressed(os.path.join(OUT, "evaluator_factors.npz"), U=Uf, V=Vf)
    json.dump(_clean(got), open(os.path.join(OUT, "evaluator_expected.json"), "w"))
    print("evaluator golden: MAP@5 =", got[5]["MAP"], "NDCG@5 =", got[5]["NDCG"])


def tiny_trajectories():
    """G2: tiny oracle trajectories (regression pins for the oracle; the GPU parity tests
    recompute the oracle live and ALSO compare against these stored tensors)."""
    from oracle.ganmf_oracle import DisGANMFOracle, GANMFOracle, reference_epoch_permutations
    rng = np.random.RandomState(2024)
    U, N = 37, 53
    urm = sps.csr_matrix((rng.rand(U, N) < 0.15).astype(np.float32))
    urm[np.arange(U), rng.randint(0, N, U)] = 1.0   # no empty rows
    urm = sps.csr_matrix(urm)
    sps.save_npz(os.path.join(OUT, "tiny_urm.npz"), urm)
    out = {}
    cases = {
        "ganmf_user_hinge_on": dict(cls="GANMF", mode="user", k=5, e=7, B=8, epochs=3, d_steps=1, g_steps=1,
                                    hp=dict(d_lr=1e-3, g_lr=2e-3, d_reg=1e-4, m=10.0, recon_coefficient=0.05)),
        "ganmf_item_steps2": dict(cls="GANMF", mode="item", k=4, e=6, B=16, epochs=2, d_steps=2, g_steps=2,
                                  hp=dict(d_lr=1e-3, g_lr=1e-3, d_reg=0.0, g_reg=1e-3, m=1.0, recon_coefficient=0.5)),
        "disganmf_user_tanh2": dict(cls="DisGANMF", mode="user", k=5, e=6, B=8, epochs=2, d_steps=1, g_steps=1,
                                    hp=dict(d_lr=1e-3, g_lr=1e-3, d_reg=1e-4, recon_coefficient=0.3,
                                            d_layers=2, d_hidden_act="tanh")),
    }
    for name, c in cases.items():
        m = urm if c["mode"] == "user" else sps.csr_matrix(urm.T)
        nu, ni = m.shape
        for prec in ("f32", "f64"):
            dt = np.float32 if prec == "f32" else np.float64
            if c["cls"] == "GANMF":
                o = GANMFOracle(nu, ni, c["k"], c["e"], dtype=dt, seed=11, **c["hp"])
            else:
                o = DisGANMFOracle(nu, ni, c["k"], d_nodes=c["e"], dtype=dt, seed=11, **c["hp"])
            if prec == "f32":
                for n, v in o.get_params().items():
                    out["%s/init/%s" % (name, n)] = v
            dls, gls = [], []
            for perm in reference_epoch_permutations(nu, c["epochs"], 1337):
                dl, gl = o.train_epoch(m, perm, c["B"], c["d_steps"], c["g_steps"])
                dls.append(dl)
                gls.append(gl)
            for n, v in o.get_params().items():
                out["%s/%s/final/%s" % (name, prec, n)] = v
            out["%s/%s/dloss" % (name, prec)] = np.concatenate(dls)
            out["%s/%s/gloss" % (name, prec)] = np.concatenate(gls)
        out["%s/config" % name] = np.array(json.dumps(c))
    np.savez_compressed(os.path.join(OUT, "tiny_trajectories.npz"), **out)
    print("tiny trajectories:", len(out), "arrays")


def statistical_fixture():
    """ML-1M train/test splits + tuned hyper-parameters + published metrics
    (test_results/GANMF_user_1M/test_results.txt:1) for the long-horizon GPU KAT."""
    for split in ("train", "test"):
        shutil.copyfile(os.path.join(REF, "experiments/datasets/Movielens1M_URM_%s.npz" % split),
                        os.path.join(OUT, "Movielens1M_URM_%s.npz" % split))
    hp = json.load(open(os.path.join(REF, "experiments/GANMF_user_1M/best_params.txt")))
    pub = pickle.load(open(os.path.join(REF, "test_results/GANMF_user_1M/test_results.pkl"), "rb"))
    json.dump({"best_params": hp, "published": _clean(pub)},
              open(os.path.join(OUT, "statistical_kat_ml1m_user.json"), "w"), indent=1)
    print("statistical fixture: published MAP@5 =", pub[5]["MAP"])
    # LastFM (BASELINE configs[0]) in both modes: the train/test splits are already fixtures (KAT-1)
    for mode in ("user", "item"):
        hp = json.load(open(os.path.join(REF, "experiments/GANMF_%s_LastFM/best_params.txt" % mode)))
        pub = pickle.load(open(os.path.join(REF, "test_results/GANMF_%s_LastFM/test_results.pkl" % mode), "rb"))
        json.dump({"best_params": hp, "published": _clean(pub)},
                  open(os.path.join(OUT, "statistical_kat_lastfm_%s.json" % mode), "w"), indent=1)
        print("statistical fixture LastFM %s: published MAP@5 =" % mode, pub[5]["MAP"])
    # hetrec2011 item mode (BASELINE configs[2])
    for split in ("train", "test"):
        shutil.copyfile(os.path.join(REF, "experiments/datasets/Movielenshetrec2011_URM_%s.npz" % split),
                        os.path.join(OUT, "hetrec2011_URM_%s.npz" % split))
    hp = json.load(open(os.path.join(REF, "experiments/GANMF_item_hetrec2011/best_params.txt")))
    pub = pickle.load(open(os.path.join(REF, "test_results/GANMF_item_hetrec2011/test_results.pkl"), "rb"))
    json.dump({"best_params": hp, "published": _clean(pub)},
              open(os.path.join(OUT, "statistical_kat_hetrec_item.json"), "w"), indent=1)
    print("statistical fixture hetrec item: published MAP@5 =", pub[5]["MAP"])
    # the remaining published GANMF rows: ML-1M item mode, hetrec2011 user mode (splits already fixtures)
    for name, exp in (("ml1m_item", "GANMF_item_1M"), ("hetrec_user", "GANMF_user_hetrec2011")):
        hp = json.load(open(os.path.join(REF, "experiments/%s/best_params.txt" % exp)))
        pub = pickle.load(open(os.path.join(REF, "test_results/%s/test_results.pkl" % exp), "rb"))
        json.dump({"best_params": hp, "published": _clean(pub)},
                  open(os.path.join(OUT, "statistical_kat_%s.json" % name), "w"), indent=1)
        print("statistical fixture %s: published MAP@5 =" % name, pub[5]["MAP"])
    # the paper's feature-matching ablation on ML-1M user mode (feature_matching/GANMF_user_1M_XX, alpha = XX/10); each
    # point has its own tuned parameters.  (The latent-factor sweep under latent_factors/ kept no parameter files.)
    abl = {}
    for kind, tags in (("feature_matching", ("00", "02", "04", "06", "08", "10")),):
        for tag in tags:
            d = os.path.join(REF, kind, "GANMF_user_1M_" + tag)
            hp = json.load(open(os.path.join(d, "best_params.txt")))
            pub = pickle.load(open(os.path.join(d, "GANMF_user_1M", "test_results.pkl"), "rb"))
            abl["%s_%s" % (kind, tag)] = {"best_params": hp, "published_map5": float(pub[5]["MAP"])}
    json.dump(abl, open(os.path.join(OUT, "statistical_kat_ml1m_user_ablations.json"), "w"), indent=1)
    print("ablation fixture:", {k: round(v["published_map5"], 4) for k, v in abl.items()})
    # DisGANMF on ML-1M (BASELINE configs[4]), both modes
    for mode in ("user", "item"):
        hp = json.load(open(os.path.join(REF, "experiments/DisGANMF_%s_1M/best_params.txt" % mode)))
        pub = pickle.load(open(os.path.join(REF, "test_results/DisGANMF_%s_1M/test_results.pkl" % mode), "rb"))
        json.dump({"best_params": hp, "published": _clean(pub)},
                  open(os.path.join(OUT, "statistical_kat_disganmf_ml1m_%s.json" % mode), "w"), indent=1)
        print("statistical fixture DisGANMF ML-1M %s: published MAP@5 =" % mode, pub[5]["MAP"])


def trial_logs():
    """The reference's own hyper-parameter search logs as known answers for the training path + the trial objective
    (RecSysExp.py:246-311): experiments/{GANMF,DisGANMF}_{user,item}_1M/results.txt hold 50 trials each as
    {fit parameters as JSON}\n{metric line of the validation evaluation}.  `epochs` in a logged line is the
    early-stop-corrected value `last_epoch - allow_worse*freq` (RecSysExp.py:272-276; 300 when early stopping never fired).
    Data only: the parameter dicts, the metrics @5 and the three splits a trial reads (train_small / early_stop /
    validation)."""
    import re
    for split in ("train_small", "early_stop", "validation"):
        shutil.copyfile(os.path.join(REF, "experiments/datasets/Movielens1M_URM_%s.npz" % split),
                        os.path.join(OUT, "Movielens1M_URM_%s.npz" % split))
    keep = ("MAP", "NDCG", "PRECISION", "RECALL", "MRR", "HIT_RATE", "ROC_AUC", "F1", "ARHR")
    out = {}
    for model in ("GANMF", "DisGANMF"):
        for mode in ("user", "item"):
            lines = open(os.path.join(REF, "experiments/%s_%s_1M/results.txt" % (model, mode))).read().split("\n")
            trials = []
            for i, line in enumerate(lines):
                if not line.startswith("{"):
                    continue
                m = re.match(r"CUTOFF: 5 - (.*)", lines[i + 1])
                metrics = dict(f.split(": ") for f in m.group(1).strip().rstrip(",").split(", "))
                trials.append({"params": json.loads(line), "validation_at5": {k: float(metrics[k]) for k in keep}})
            assert len(trials) == 50
            best = json.load(open(os.path.join(REF, "experiments/%s_%s_1M/best_params.txt" % (model, mode))))
            out["%s_%s_1M" % (model, mode)] = {"trials": trials, "best_params": best}
            print("trial log %s %s: 50 trials, logged MAP@5 in [%.4f, %.4f]" % (
                model, mode, min(t["validation_at5"]["MAP"] for t in trials), max(t["validation_at5"]["MAP"] for t in trials)))
    json.dump({"source": "experiments/<model>_<mode>_1M/results.txt", "early_stopping": {"allow_worse": 5, "freq": 5},
               "metric": "MAP", "at": 5, "experiments": out},
              open(os.path.join(OUT, "trial_logs_ml1m.json"), "w"), indent=0)


def mf_contract_golden():
    """MF-contract corners of _compute_item_score / recommend (Base/BaseMatrixFactorizationRecommender.py:113-119,
    128-143): items_to_compute -> every other item scores -inf; users without a training interaction ("cold") score -inf
    everywhere.  Expected outputs come from the reference's own class on seeded random factors; the fixture holds inputs and
    outputs only."""
    sys.path.insert(0, REF)
    from Base.BaseMatrixFactorizationRecommender import BaseMatrixFactorizationRecommender
    rng = np.random.RandomState(20251004)
    n_users, n_items, k = 61, 83, 7
    dense = (rng.rand(n_users, n_items) < 0.12).astype(np.float32)
    cold = np.array([3, 17, 40])
    dense[cold] = 0.0
    dense[np.setdiff1d(np.arange(n_users), cold), rng.randint(0, n_items, n_users - len(cold))] = 1.0      # everybody else is warm
    urm = sps.csr_matrix(dense)
    U = rng.randn(n_users, k).astype(np.float32)
    V = rng.randn(n_items, k).astype(np.float32)

    class RefMF(BaseMatrixFactorizationRecommender):
        RECOMMENDER_NAME = "mf_contract"

    rec = RefMF(urm)
    rec.USER_factors, rec.ITEM_factors = U, V
    users = np.array([0, 3, 5, 17, 18, 40, 41, 60], dtype=np.int64)
    items = np.sort(rng.choice(n_items, 29, replace=False)).astype(np.int64)
    out = {"users": users, "items_to_compute": items, "cold_users": cold, "U": U, "V": V,
           "urm_indptr": urm.indptr.astype(np.int64), "urm_indices": urm.indices.astype(np.int32), "urm_shape": np.array(urm.shape)}
    out["scores_all"] = rec._compute_item_score(users)
    out["scores_subset"] = rec._compute_item_score(users, items_to_compute=items)
    for name, kw in (("rank_all_seen", dict(remove_seen_flag=True)), ("rank_subset_seen", dict(remove_seen_flag=True, items_to_compute=items)),
                     ("rank_subset_unseen", dict(remove_seen_flag=False, items_to_compute=items))):
        lists = rec.recommend(users, cutoff=10, **kw)
        pad = np.full((len(users), 10), -1, dtype=np.int32)
        for i, l in enumerate(lists):
            pad[i, :len(l)] = l
        out[name] = pad

    # The reference's OWN GANMF contract on the same matrix and factors (GANRec/GANMF.py:285-292): the class derives from
    # BaseRecommender (no cold-user mask), `_compute_item_score` is the plain product U[ids] . V^T for every user and takes
    # `items_to_compute` without reading it.  The product is restated here (the TF session cannot run); everything around it --
    # seen-item removal, partition / sort order, -inf filtering of the lists -- is the reference's own BaseRecommender.recommend.
    from Base.BaseRecommender import BaseRecommender

    class RefGANMFScores(BaseRecommender):
        RECOMMENDER_NAME = "ganmf_contract"

        def _compute_item_score(self, user_id_array, items_to_compute=None):
            return np.dot(U[np.asarray(user_id_array).reshape(-1)], V.T)

    rec = RefGANMFScores(urm)
    out["ganmf_scores_all"] = rec._compute_item_score(users)
    for name, kw in (("ganmf_rank_all_seen", dict(remove_seen_flag=True)), ("ganmf_rank_subset_seen", dict(remove_seen_flag=True, items_to_compute=items)),
                     ("ganmf_rank_all_unseen", dict(remove_seen_flag=False))):
        lists = rec.recommend(users, cutoff=10, **kw)
        assert all(len(l) == 10 for l in lists)      # every user, cold or not, is recommended items
        out[name] = np.array(lists, dtype=np.int32)
    assert np.array_equal(out["ganmf_rank_all_seen"], out["ganmf_rank_subset_seen"])      # items_to_compute changes nothing
    np.savez_compressed(os.path.join(OUT, "mf_contract.npz"), **out)
    print("MF contract fixture: %d users (%d cold), %d of %d items to compute" % (len(users), len(cold), len(items), n_items))


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    BR, EH = _import_reference()
    if len(sys.argv) > 1 and sys.argv[1] == "mf_contract":      # one fixture only
        mf_contract_golden()
        raise SystemExit(0)
    kat1(BR, EH)
    bundle_golden()
    evaluator_golden(BR, EH)
    tiny_trajectories()
    statistical_fixture()
    trial_logs()
    mf_contract_golden()
