"""Replays the reference's logged hyper-parameter trials (tests/golden/trial_logs_ml1m.json, copied from
experiments/<model>_<mode>_1M/results.txt by oracle/make_golden.py) through this build's trial objective
(ganmf_amd.tune.run_trial = RecSysExp.obj_func, RecSysExp.py:246-311): fit on URM_train_small for up to 300 epochs with
the GAN early-stopping dict on URM_early_stop, MAP@5 on URM_validation, `epochs` corrected the way the reference logs it.

    python tools/replay_trials.py [--experiments GANMF_user_1M,DisGANMF_user_1M] [--trials 0-49] [--seeds 1337] [--out file.json]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import scipy.sparse as sps

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def load_fixture():
    logs = json.load(open(os.path.join(GOLDEN, "trial_logs_ml1m.json")))
    splits = {s: sps.load_npz(os.path.join(GOLDEN, "Movielens1M_URM_%s.npz" % s)).tocsr()
              for s in ("train_small", "early_stop", "validation")}
    return logs, splits


def replay(experiment, trial_ids, seed=1337, device=0, logs=None, splits=None, verbose=True, concurrency=1):
    """-> list of dicts {trial, logged_map, map, logged_epochs, epochs, seconds} for one experiment.
    concurrency > 1: that many trials at a time in threads of this process, each with its own engine (HIP stream) and its own
    RandomState(seed) for the epoch shuffles -- the numbers numpy's global stream would draw after np.random.seed(seed), so
    every trial computes exactly what it computes alone (tests/test_gpu_recommend.py::test_concurrent_engines_bit_identical)."""
    from GANRec.DisGANMF import DisGANMF
    from GANRec.GANMF import GANMF
    from ganmf_amd import tune
    from ganmf_amd.evaluation import EvaluatorHoldoutFast
    if logs is None:
        logs, splits = load_fixture()
    model_name, mode, _ = experiment.split("_")
    cls = {"GANMF": GANMF, "DisGANMF": DisGANMF}[model_name]
    spec = {"recommender_class": cls, "URM_train_small": splits["train_small"], "URM_early_stop": splits["early_stop"],
            "URM_validation": splits["validation"], "mode": mode, "seed": seed, "metric": logs["metric"], "at": logs["at"],
            "evaluator_class": EvaluatorHoldoutFast, "model_kwargs": {}, "visible_devices": None}
    def one(i):
        logged = logs["experiments"][experiment]["trials"][i]
        params = dict(logged["params"])
        logged_epochs = params["epochs"]
        params["epochs"] = 300                      # every GAN trial starts from the one-point `epochs` dimension (RecSysExp.py:502-523)
        t0 = time.time()
        # the minibatch schedule draws from a stream seeded per trial (= numpy's global stream after np.random.seed(seed))
        res = tune.run_trial(spec, params, device, schedule_rng=np.random.RandomState(seed))
        rec = {"trial": i, "logged_map": logged["validation_at5"]["MAP"], "map": -res["fitness"],
               "logged_epochs": logged_epochs, "epochs": res["fit_params"]["epochs"], "seconds": time.time() - t0}
        if verbose:
            print("%s trial %2d: MAP@5 %.4f (logged %.4f)  epochs %3d (logged %3d)  %.1f s  %s" % (
                experiment, i, rec["map"], rec["logged_map"], rec["epochs"], logged_epochs, rec["seconds"],
                {k: v for k, v in logged["params"].items() if k in ("d_hidden_act", "d_layers", "batch_size", "num_factors")}),
                flush=True)
        return rec

    trial_ids = list(trial_ids)
    if concurrency <= 1:
        return [one(i) for i in trial_ids]
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=concurrency) as pool:
        return list(pool.map(one, trial_ids))


def spearman(a, b):
    from scipy.stats import spearmanr
    return float(spearmanr(a, b)[0])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--experiments", default="GANMF_user_1M,DisGANMF_user_1M")
    ap.add_argument("--trials", default="0-49")
    ap.add_argument("--seeds", default="1337")
    ap.add_argument("--out", default=None)
    ap.add_argument("--concurrency", type=int, default=1)
    a = ap.parse_args()
    lo, hi = (a.trials.split("-") + [a.trials])[:2] if "-" in a.trials else (a.trials, a.trials)
    ids = list(range(int(lo), int(hi) + 1))
    logs, splits = load_fixture()
    summary = {}
    for exp in a.experiments.split(","):
        for seed in [int(s) for s in a.seeds.split(",")]:
            t_exp = time.time()
            rows = replay(exp, ids, seed=seed, logs=logs, splits=splits, concurrency=a.concurrency)
            print("%s seed %d: %d trials in %.1f s wall at concurrency %d" % (exp, seed, len(rows), time.time() - t_exp, a.concurrency), flush=True)
            rho = spearman([r["logged_map"] for r in rows], [r["map"] for r in rows]) if len(rows) > 2 else float("nan")
            ok = sum(abs(r["map"] - r["logged_map"]) <= max(0.01, 0.15 * r["logged_map"]) for r in rows)
            print("%s seed %d: Spearman rho %.3f over %d trials; %d within max(0.01, 15%%); %.1f s" % (
                exp, seed, rho, len(rows), ok, sum(r["seconds"] for r in rows)), flush=True)
            summary["%s/seed%d" % (exp, seed)] = {"spearman": rho, "within_band": ok, "rows": rows}
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        json.dump(summary, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
