"""Long-horizon parity: full training with the reference's tuned hyper-parameters on the
reference's own MovieLens-1M split must land on the published accuracy
(test_results/GANMF_user_1M/test_results.txt:1; SURVEY F12: a +-0.005 band on MAP@5 / NDCG@5 is
the realistic acceptance criterion because TF's seeded Glorot init is not reproducible)."""
import json
import os
import time

import numpy as np
import pytest
import scipy.sparse as sps

pytestmark = [pytest.mark.gpu, pytest.mark.slow]


def test_ml1m_user_full_training_reaches_published_map(golden_dir):
    from ganmf_amd.GANMF import GANMF
    from ganmf_amd.evaluation import EvaluatorHoldout
    kat = json.load(open(os.path.join(golden_dir, "statistical_kat_ml1m_user.json")))
    train = sps.load_npz(os.path.join(golden_dir, "Movielens1M_URM_train.npz")).tocsr()
    test = sps.load_npz(os.path.join(golden_dir, "Movielens1M_URM_test.npz")).tocsr()
    np.random.seed(1337)                                  # RecSysExp.set_seed (RecSysExp.py:104-108)
    model = GANMF(train, mode='user', seed=1337, is_experiment=True)
    t0 = time.time()
    ret = model.fit(validation_set=None, sample_every=None, validation_evaluator=None, **kat["best_params"])
    train_s = time.time() - t0
    assert ret == kat["best_params"]["epochs"] + 1
    res, _ = EvaluatorHoldout(test, [5, 10, 20, 50]).evaluateRecommender(model)
    pub = kat["published"]
    steps = kat["best_params"]["epochs"] * 2 * -(-train.shape[0] // kat["best_params"]["batch_size"])
    print("ML-1M GANMF-user: %d updates in %.2f s (%.0f steps/s); MAP@5 %.4f (published %.4f) NDCG@5 %.4f (%.4f)"
          % (steps, train_s, steps / train_s, res[5]["MAP"], pub["5"]["MAP"], res[5]["NDCG"], pub["5"]["NDCG"]))
    for metric in ("MAP", "NDCG", "PRECISION", "RECALL"):
        assert abs(res[5][metric] - pub["5"][metric]) <= 0.005, (metric, res[5][metric], pub["5"][metric])
    for c in ("10", "20", "50"):
        assert abs(res[int(c)]["MAP"] - pub[c]["MAP"]) <= 0.005, (c, res[int(c)]["MAP"], pub[c]["MAP"])


DATASETS = {"lastfm_user": ("LastFM", "user"), "lastfm_item": ("LastFM", "item"), "hetrec_item": ("hetrec2011", "item"),
            "hetrec_user": ("hetrec2011", "user"), "ml1m_item": ("Movielens1M", "item")}


@pytest.mark.parametrize("case", list(DATASETS))
def test_lastfm_hetrec_full_training_reaches_published_map(golden_dir, case):
    """BASELINE configs[0] (LastFM 1892 x 17632, user and item mode) and configs[2] (hetrec2011 item mode,
    2113 x 10109) with the reference's tuned hyper-parameters (experiments/GANMF_*/best_params.txt) against
    test_results/GANMF_*/test_results.txt:1.  Same +-0.005 band; the device-side recommend path and the block
    evaluator produce the metrics."""
    from ganmf_amd.GANMF import GANMF
    from ganmf_amd.evaluation import EvaluatorHoldoutFast
    data, mode = DATASETS[case]
    kat = json.load(open(os.path.join(golden_dir, "statistical_kat_%s.json" % case)))
    train = sps.load_npz(os.path.join(golden_dir, "%s_URM_train.npz" % data)).tocsr()
    test = sps.load_npz(os.path.join(golden_dir, "%s_URM_test.npz" % data)).tocsr()
    np.random.seed(1337)
    model = GANMF(train, mode=mode, seed=1337, is_experiment=True)
    t0 = time.time()
    ret = model.fit(validation_set=None, sample_every=None, validation_evaluator=None, **kat["best_params"])
    train_s = time.time() - t0
    assert ret == kat["best_params"]["epochs"] + 1
    res, _ = EvaluatorHoldoutFast(test, [5, 10]).evaluateRecommender(model)
    pub = kat["published"]
    rows = model.num_users
    steps = kat["best_params"]["epochs"] * 2 * -(-rows // kat["best_params"]["batch_size"])
    print("%s GANMF-%s: %d updates in %.2f s (%.0f steps/s); MAP@5 %.4f (published %.4f) NDCG@5 %.4f (%.4f)"
          % (data, mode, steps, train_s, steps / train_s, res[5]["MAP"], pub["5"]["MAP"], res[5]["NDCG"], pub["5"]["NDCG"]))
    for metric in ("MAP", "NDCG", "PRECISION", "RECALL"):
        assert abs(res[5][metric] - pub["5"][metric]) <= 0.005, (case, metric, res[5][metric], pub["5"][metric])
    assert abs(res[10]["MAP"] - pub["10"]["MAP"]) <= 0.005


@pytest.mark.parametrize("mode", ["user", "item"])
def test_disganmf_ml1m_full_training(golden_dir, mode):
    """BASELINE configs[4]: DisGANMF on ML-1M with the reference's tuned hyper-parameters
    (experiments/DisGANMF_{user,item}_1M/best_params.txt) vs test_results/DisGANMF_*_1M/test_results.txt:1, as a
    DISTRIBUTION over eight initialisations instead of one seed inside a wide band.

    The binary-discriminator GAN is far more init-sensitive than GANMF (its generator *minimises* loss_fake as the
    reference writes it, DisGANMF.py:132-136; the raw float(uid) column drives layer 0 with inputs up to 6040).  The
    numpy oracle shows the same spread (oracle/run_end_to_end.py: 0.1353 at seed 1337; SURVEY F12's probe 0.1509 with
    another init stream).  The published row is ONE run of the configuration that won a 50-trial search, so it is
    expected in the upper part of the distribution, not at its mean:
      * the published MAP@5 must lie inside [min - 0.012, max + 0.012] of the eight runs;
      * the mean may sit below it by no more than 0.025 and above it by no more than 0.01.
    What pins the DisGANMF arithmetic tightly is tests/test_gpu_trial_logs.py: the reference's 100 logged DisGANMF
    trials (all four activations, 1-5 layers) replayed with rho 0.83-0.89 and the mean MAP@5 within 3 %."""
    from ganmf_amd.DisGANMF import DisGANMF
    from ganmf_amd.evaluation import EvaluatorHoldoutFast
    kat = json.load(open(os.path.join(golden_dir, "statistical_kat_disganmf_ml1m_%s.json" % mode)))
    train = sps.load_npz(os.path.join(golden_dir, "Movielens1M_URM_train.npz")).tocsr()
    test = sps.load_npz(os.path.join(golden_dir, "Movielens1M_URM_test.npz")).tocsr()
    ev = EvaluatorHoldoutFast(test, [5])
    pub = kat["published"]["5"]["MAP"]
    vals = []
    for seed in (1337, 1, 2, 3, 4, 5, 6, 7):
        np.random.seed(seed)
        model = DisGANMF(train, mode=mode, seed=seed, is_experiment=True)
        model.fit(validation_set=None, sample_every=None, validation_evaluator=None, **kat["best_params"])
        vals.append(ev.evaluateRecommender(model)[0][5]["MAP"])
        model.engine.close()
    vals = np.array(vals)
    print("ML-1M DisGANMF-%s MAP@5 over 8 seeds: min %.4f mean %.4f max %.4f (published %.4f)  %s"
          % (mode, vals.min(), vals.mean(), vals.max(), pub, np.round(vals, 4)))
    assert vals.min() - 0.012 <= pub <= vals.max() + 0.012, (vals, pub)
    assert -0.025 <= vals.mean() - pub <= 0.01, (vals.mean(), pub)


@pytest.mark.parametrize("case", ["ganmf_ml1m_user", "ganmf_hetrec_item", "ganmf_lastfm_user"])
def test_hip_matches_oracle_end_to_end(golden_dir, case):
    """The HIP path against the numpy oracle's OWN full training run (oracle/run_end_to_end.py, committed as
    tests/golden/oracle_end_to_end.json): same hyper-parameters, same initial weights (RandomState(1337) draws in the
    same tensor order), same minibatch schedule.  Per-step agreement is tested elsewhere at 1e-4; over 10^4 chaotic
    updates the two fp32 trajectories separate, so the end-to-end comparison is on the metrics and the factor norms:
    GANMF within 0.004 on every metric @5 and on MAP@10/20/50 (the published row sits inside the same band) and 1 % on the
    factor norms.  DisGANMF is NOT held to its oracle run end to end: over 10^4 updates its trajectory is chaotic enough that
    two fp32 implementations with different summation orders are two draws of one distribution (tools/chaos_check.py), and
    a band as wide as that distribution cannot fail.  Its arithmetic is pinned per step (tests/test_gpu_disganmf.py), over
    200 updates at the configs[4] shape with a stated tolerance per horizon (tests/test_gpu_trajectory.py), by the
    reference's 100 logged trials (tests/test_gpu_trial_logs.py) and as a distribution over initialisations
    (test_disganmf_ml1m_full_training above); the oracle's own DisGANMF run is held to the published row and the measured
    spread in tests/test_oracle_pin.py."""
    fx = json.load(open(os.path.join(golden_dir, "oracle_end_to_end.json")))
    if case not in fx:
        pytest.skip("oracle run %s not committed" % case)
    o = fx[case]
    from ganmf_amd.DisGANMF import DisGANMF
    from ganmf_amd.GANMF import GANMF
    from ganmf_amd.evaluation import EvaluatorHoldoutFast
    train = sps.load_npz(os.path.join(golden_dir, "%s_URM_train.npz" % o["dataset"])).tocsr()
    test = sps.load_npz(os.path.join(golden_dir, "%s_URM_test.npz" % o["dataset"])).tocsr()
    np.random.seed(o["seed"])
    cls = GANMF if o["model"] == "GANMF" else DisGANMF
    model = cls(train, mode=o["mode"], seed=o["seed"], is_experiment=True)
    model.fit(validation_set=None, sample_every=None, validation_evaluator=None, **o["best_params"])
    res, _ = EvaluatorHoldoutFast(test, [5, 10, 20, 50]).evaluateRecommender(model)
    assert o["model"] == "GANMF"
    tol = 0.004
    ref = o["oracle_metrics"]
    print("%s: HIP MAP@5 %.4f NDCG@5 %.4f | oracle %.4f %.4f | published %.4f %.4f" % (
        case, res[5]["MAP"], res[5]["NDCG"], ref["5"]["MAP"], ref["5"]["NDCG"], o["published_at5"]["MAP"], o["published_at5"]["NDCG"]))
    for metric in ("MAP", "NDCG", "PRECISION", "RECALL"):
        assert abs(res[5][metric] - ref["5"][metric]) <= tol, (metric, res[5][metric], ref["5"][metric])
    for c in (10, 20, 50):
        assert abs(res[c]["MAP"] - ref[str(c)]["MAP"]) <= tol, (c, res[c]["MAP"], ref[str(c)]["MAP"])
    norm_tol = 0.01
    for name, got in (("U", model.user_factors()), ("V", model.item_factors())):
        want = o["factor_norms"][name]
        assert abs(np.linalg.norm(got.astype(np.float64)) - want) <= norm_tol * want, (name, np.linalg.norm(got), want)
    model.engine.close()


def test_ml1m_user_feature_matching_ablation(golden_dir):
    """The paper's feature-matching ablation on ML-1M user mode, every point trained with the parameters the reference
    tuned for it: alpha in {0, .2, .4, .6, .8, 1} (feature_matching/GANMF_user_1M_*).  +-0.01 on MAP@5 per point at seed 1337
    and the published shape of the curve: alpha = 0 (no feature matching) collapses to less than 60 % of any other point.
    The alpha = 0.2 point (20 epochs at d_lr 2.4e-3, g_lr 1.5e-3: the largest learning rates of the six) is bimodal across
    initialisations, so it is tested as a DISTRIBUTION over eight seeds, like DisGANMF above: the published 0.3474 (one run
    of the configuration that won the search: expected in the upper mode) must lie inside [min - 0.01, max + 0.01] of the
    eight runs, and at least two of the eight must land within 0.01 of it (an upper mode exists; it is not one lucky seed)."""
    from ganmf_amd.GANMF import GANMF
    from ganmf_amd.evaluation import EvaluatorHoldoutFast
    abl = json.load(open(os.path.join(golden_dir, "statistical_kat_ml1m_user_ablations.json")))
    train = sps.load_npz(os.path.join(golden_dir, "Movielens1M_URM_train.npz")).tocsr()
    test = sps.load_npz(os.path.join(golden_dir, "Movielens1M_URM_test.npz")).tocsr()
    ev = EvaluatorHoldoutFast(test, [5])

    def run(point, seed):
        np.random.seed(seed)
        model = GANMF(train, mode="user", seed=seed, is_experiment=True)
        model.fit(validation_set=None, sample_every=None, validation_evaluator=None, **point["best_params"])
        v = ev.evaluateRecommender(model)[0][5]["MAP"]
        model.engine.close()
        return v
    got = {}
    for name, point in abl.items():
        if name == "feature_matching_02":
            continue
        got[name] = run(point, 1337)
        print("%-24s MAP@5 %.4f (published %.4f)" % (name, got[name], point["published_map5"]))
        assert abs(got[name] - point["published_map5"]) <= 0.01, (name, got[name], point["published_map5"])
    p02 = abl["feature_matching_02"]
    vals = np.array([run(p02, seed) for seed in (1337, 1, 2, 3, 4, 5, 6, 7)])
    pub = p02["published_map5"]
    print("feature_matching_02 over 8 seeds: %s  (published %.4f)" % (np.round(vals, 4), pub))
    assert vals.min() - 0.01 <= pub <= vals.max() + 0.01, (vals, pub)
    assert np.sum(np.abs(vals - pub) <= 0.01) >= 2, (vals, pub)
    assert got["feature_matching_00"] < 0.6 * min(min(v for k, v in got.items() if not k.endswith("_00")), vals.max())
