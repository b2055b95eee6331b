"""The reference's own hyper-parameter search logs as known answers for TRAINING (GANMF a1-a14, DisGANMF a18-a20) and
for the trial objective: experiments/{GANMF,DisGANMF}_{user,item}_1M/results.txt hold 50 trials each --
fit parameters (all four activations, d_layers 1-5, batch 64-1024, 1-250 factors) and the validation metrics the
reference measured for them.  Every trial is replayed through this build's objective (ganmf_amd.tune.run_trial =
RecSysExp.obj_func, RecSysExp.py:246-311): fit on URM_train_small for up to 300 epochs with early stopping on
URM_early_stop, MAP@5 on URM_validation.

TensorFlow's seeded Glorot initialisation cannot be reproduced, GAN training is chaotic and the epoch
early stopping fires at moves with the initialisation, so single trials scatter; what a faithful implementation must
reproduce is the RESPONSE to the hyper-parameters.  Measured on MI355X over seeds 1337 / 1 / 2 (profiles/
r02_trial_replay.md): Spearman rho 0.93-0.96 (GANMF user), 0.94-0.97 (GANMF item), 0.83-0.89 (DisGANMF user),
0.88 (DisGANMF item); 40-44 / 42-44 / 36-40 / 35 of 50 trials within max(0.01, 15 %) of the logged MAP@5; the mean MAP@5
over the 50 trials within 1 % of the logged mean (DisGANMF item: 3 %) and the mean early-stopping epoch within 5 %
(DisGANMF item: 14 %).  The assertions below leave room for seed-to-seed variation of those figures, not more."""
import json
import os
import sys

import numpy as np
import pytest

pytestmark = [pytest.mark.gpu, pytest.mark.slow]

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

# experiment -> (min Spearman rho, min trials in band at the first seed, max trials never in band over three seeds,
#                max relative error of the mean MAP@5 over the 50 trials)
EXPECT = {
    "GANMF_user_1M": (0.88, 37, 8, 0.05),
    "GANMF_item_1M": (0.88, 37, 8, 0.05),
    "DisGANMF_user_1M": (0.78, 33, 9, 0.08),
    "DisGANMF_item_1M": (0.78, 31, 12, 0.10),
}


CONCURRENCY = 3


def _in_band(got, logged):
    return abs(got - logged) <= max(0.01, 0.15 * logged)


@pytest.mark.parametrize("experiment", list(EXPECT))
def test_logged_trials_replay(experiment):
    import replay_trials as R
    rho_min, band_min, never_max, mean_tol = EXPECT[experiment]
    logs, splits = R.load_fixture()
    # three trials at a time in threads of this process (own engine, own stream and own RandomState(seed) each): every trial
    # computes exactly what it computes alone (test_concurrent_engines_bit_identical), the 50 take ~30 % less wall time
    rows = R.replay(experiment, range(50), seed=1337, logs=logs, splits=splits, verbose=False, concurrency=CONCURRENCY)
    logged = np.array([r["logged_map"] for r in rows])
    got = np.array([r["map"] for r in rows])
    rho = R.spearman(logged, got)
    ok = np.array([_in_band(g, l) for g, l in zip(got, logged)])
    print("%s: Spearman rho %.3f, %d / 50 within max(0.01, 15%%), mean MAP@5 %.4f (logged %.4f), mean epochs %.1f (logged %.1f)"
          % (experiment, rho, ok.sum(), got.mean(), logged.mean(), np.mean([r["epochs"] for r in rows]),
             np.mean([r["logged_epochs"] for r in rows])))
    assert rho >= rho_min, rho
    assert ok.sum() >= band_min, int(ok.sum())
    assert abs(got.mean() - logged.mean()) <= mean_tol * logged.mean(), (got.mean(), logged.mean())
    # activations / depths the single published DisGANMF row does not reach (linear, 1 layer): every family of the
    # search space must respond like the reference
    if experiment.startswith("DisGANMF"):
        params = [logs["experiments"][experiment]["trials"][i]["params"] for i in range(50)]
        for act in ("linear", "tanh", "relu", "sigmoid"):
            sel = np.array([p["d_hidden_act"] == act for p in params])
            if sel.sum() < 3:       # a one- or two-trial "family" (relu: one logged trial) is a single chaotic GAN run, not a
                continue            # mean: such trials are held to the per-trial rule with second seeds below
            assert abs(got[sel].mean() - logged[sel].mean()) <= max(0.015, 0.3 * logged[sel].mean()), (act, got[sel].mean(), logged[sel].mean())
        deep = np.array([p["d_layers"] >= 3 for p in params])
        assert abs(got[deep].mean() - logged[deep].mean()) <= max(0.01, 0.2 * logged[deep].mean())
    # second chances: a trial outside the band is re-run with two other seeds; few may stay outside
    never = 0
    out = [int(i) for i in np.where(~ok)[0]]
    second = {s: {r["trial"]: r["map"] for r in R.replay(experiment, out, seed=s, logs=logs, splits=splits, verbose=False,
                                                         concurrency=CONCURRENCY)} for s in (1, 2)} if out else {}
    for i in out:
        again = [second[s][i] for s in (1, 2)]
        lo, hi = min(again + [got[i]]), max(again + [got[i]])
        band = max(0.01, 0.15 * logged[i])
        if not (lo - band <= logged[i] <= hi + band):
            never += 1
    print("%s: %d trials outside [min - band, max + band] over seeds 1337 / 1 / 2" % (experiment, never))
    assert never <= never_max, never
