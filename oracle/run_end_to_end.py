"""The oracle's OWN end-to-end training runs (SURVEY F12 made reproducible).  TEST INFRASTRUCTURE ONLY.

Trains the float32 numpy oracle (oracle/ganmf_oracle.py) with the reference's tuned hyper-parameters
(experiments/<model>_<mode>_<dataset>/best_params.txt, copied into tests/golden/statistical_kat_*.json by
make_golden.py) on the reference's own train split, with the reference's minibatch schedule (np.random.seed(1337),
one cumulative in-place shuffle per epoch, GANMF.py:175), scores the test split and stores the metrics next to the
published row of test_results/<...>/test_results.pkl in tests/golden/oracle_end_to_end.json.

This is what ties the CPU restatement to the reference's published numbers independently of the HIP path; the GPU
test tests/test_gpu_statistical.py::test_hip_matches_oracle_end_to_end then compares the HIP path with THESE numbers
(same initial weights, same schedule) at a much tighter band than with the published row.

    python oracle/run_end_to_end.py [case ...]        # cases: ganmf_ml1m_user disganmf_ml1m_user ganmf_hetrec_item ganmf_lastfm_user
Runs minutes per case on 8 host cores (12 540 / 10 080 / 17 380 / 404 minibatch updates)."""
import json
import os
import sys
import time

import numpy as np
import scipy.sparse as sps

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
OUT = os.path.join(GOLDEN, "oracle_end_to_end.json")

CASES = {
    "ganmf_ml1m_user": ("GANMF", "user", "Movielens1M", "statistical_kat_ml1m_user.json"),
    "disganmf_ml1m_user": ("DisGANMF", "user", "Movielens1M", "statistical_kat_disganmf_ml1m_user.json"),
    "ganmf_hetrec_item": ("GANMF", "item", "hetrec2011", "statistical_kat_hetrec_item.json"),
    "ganmf_lastfm_user": ("GANMF", "user", "LastFM", "statistical_kat_lastfm_user.json"),
}
SEED = 1337


def run_case(name):
    from ganmf_amd.base import BaseRecommender           # host-side ranking / metrics (pinned to the reference's
    from ganmf_amd.evaluation import EvaluatorHoldout    # golden outputs by tests/test_evaluator.py); no HIP involved
    from oracle.ganmf_oracle import DisGANMFOracle, GANMFOracle, reference_epoch_permutations
    model, mode, dataset, kat = CASES[name]
    fx = json.load(open(os.path.join(GOLDEN, kat)))
    hp = dict(fx["best_params"])
    train = sps.load_npz(os.path.join(GOLDEN, "%s_URM_train.npz" % dataset)).tocsr().astype(np.float32)
    test = sps.load_npz(os.path.join(GOLDEN, "%s_URM_test.npz" % dataset)).tocsr()
    fit_urm = sps.csr_matrix(train.T) if mode == "item" else train
    nu, ni = fit_urm.shape
    epochs, B = hp.pop("epochs"), hp.pop("batch_size")
    if model == "GANMF":
        o = GANMFOracle(nu, ni, hp["num_factors"], hp["emb_dim"], d_lr=hp["d_lr"], g_lr=hp["g_lr"], d_reg=hp["d_reg"],
                        g_reg=hp.get("g_reg", 0.0), m=hp["m"], recon_coefficient=hp["recon_coefficient"],
                        dtype=np.float32, seed=SEED)
    else:
        o = DisGANMFOracle(nu, ni, hp["num_factors"], d_layers=hp["d_layers"], d_nodes=hp["d_nodes"],
                           d_hidden_act=hp["d_hidden_act"], d_lr=hp["d_lr"], g_lr=hp["g_lr"], d_reg=hp["d_reg"],
                           g_reg=hp.get("g_reg", 0.0), recon_coefficient=hp["recon_coefficient"], dtype=np.float32,
                           seed=SEED)
    t0 = time.time()
    n_updates = 0
    last_d = last_g = float("nan")
    for ep, perm in enumerate(reference_epoch_permutations(nu, epochs, SEED), 1):
        dl, gl = o.train_epoch(fit_urm, perm, B)
        n_updates += len(dl) + len(gl)
        last_d, last_g = float(np.mean(dl)), float(np.mean(gl))
        if ep % 10 == 0 or ep == epochs:
            print("%s epoch %d/%d  dloss %.6f gloss %.6f  %.0f s" % (name, ep, epochs, last_d, last_g, time.time() - t0), flush=True)
    seconds = time.time() - t0

    class Scorer(BaseRecommender):
        def _compute_item_score(self, user_id_array, items_to_compute=None):
            return o.scores(np.asarray(user_id_array), item_mode=(mode == "item")).astype(np.float32)

    res, _ = EvaluatorHoldout(test, [5, 10, 20, 50]).evaluateRecommender(Scorer(train))
    p = o.get_params()
    probe = np.arange(0, train.shape[0], max(1, train.shape[0] // 8))[:8]
    return {
        "model": model, "mode": mode, "dataset": dataset, "seed": SEED, "best_params": fx["best_params"],
        "updates": n_updates, "seconds_on_host": seconds, "final_dloss": last_d, "final_gloss": last_g,
        "oracle_metrics": {str(c): {k: float(v) for k, v in res[c].items()} for c in res},
        "published_at5": fx["published"]["5"],
        "factor_norms": {"U": float(np.linalg.norm(p["U"])), "V": float(np.linalg.norm(p["V"]))},
        "probe_users": probe.tolist(),
        "probe_score_row_norms": [float(np.linalg.norm(r)) for r in
                                  o.scores(probe, item_mode=(mode == "item")).astype(np.float64)],
    }


def main():
    names = sys.argv[1:] or list(CASES)
    data = json.load(open(OUT)) if os.path.exists(OUT) else {}
    for n in names:
        data[n] = run_case(n)
        m = data[n]["oracle_metrics"]["5"]
        print("%s: oracle MAP@5 %.4f NDCG@5 %.4f (published %.4f / %.4f), %d updates in %.0f s" % (
            n, m["MAP"], m["NDCG"], data[n]["published_at5"]["MAP"], data[n]["published_at5"]["NDCG"],
            data[n]["updates"], data[n]["seconds_on_host"]), flush=True)
        # merge with what a concurrent run of another case may have written meanwhile
        cur = json.load(open(OUT)) if os.path.exists(OUT) else {}
        cur[n] = data[n]
        json.dump(cur, open(OUT, "w"), indent=1)


if __name__ == "__main__":
    main()
