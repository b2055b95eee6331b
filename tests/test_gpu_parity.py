"""HIP path vs the CPU oracle on identical initial weights and batch schedules, through the C ABI.

Tolerances (fp32 arithmetic, different summation orders; SURVEY §7 'Parity under Adam'):
  parameters / scores after T updates are compared with the fp64 oracle as
      |delta| <= RTOL_T * max|reference tensor|          (normalised by the tensor's scale)
  with RTOL = 2e-5 for T <= 10 and 1e-4 for T <= 200 (north star: scores within 1e-4 rel).
"""
import json
import os

import numpy as np
import pytest
import scipy.sparse as sps

from oracle.ganmf_oracle import GANMFOracle, batch_slices, reference_epoch_permutations

pytestmark = pytest.mark.gpu

NAME2ID = {"We": 0, "be": 1, "Wd": 2, "bd": 3, "U": 100, "V": 101}


def _rand_urm(rng, U, N, density):
    m = (rng.rand(U, N) < density).astype(np.float32)
    m[np.arange(U), rng.randint(0, N, U)] = 1.0
    return sps.csr_matrix(m)


def _engine_from_oracle(o, urm, B, hp):
    from ganmf_amd.engine import Engine
    eng = Engine(o.nu, o.ni, o.k, o.e, B, **hp)
    eng.set_urm(urm)
    for n, tid in NAME2ID.items():
        eng.set_tensor(tid, o.p[n])
    return eng


def _get(eng, name):
    a = eng.get_tensor(NAME2ID[name])
    return a[0] if name in ("be", "bd") else a


def _close(got, ref, rtol, what):
    scale = np.max(np.abs(ref)) + 1e-30
    err = np.max(np.abs(got.astype(np.float64) - ref.astype(np.float64))) / scale
    assert err <= rtol, "%s: normalised error %.3e > %.1e" % (what, err, rtol)
    return err


HP = dict(d_lr=1e-3, g_lr=2e-3, d_reg=1e-4, g_reg=0.0, m=10.0, recon_coefficient=0.05)


@pytest.mark.parametrize("shape", [(37, 53, 5, 7, 8), (300, 517, 33, 65, 64), (700, 3706, 250, 992, 128)])
@pytest.mark.parametrize("m", [10.0, 0.001])   # hinge active / inactive
def test_single_steps_match_oracle(shape, m):
    U, N, k, e, B = shape
    rng = np.random.RandomState(U + N)
    urm = _rand_urm(rng, U, N, 0.05)
    hp = dict(HP, m=m)
    o = GANMFOracle(U, N, k, e, dtype=np.float64, seed=5, **hp)
    o.set_params(be=rng.randn(e) * 0.01, bd=rng.randn(N) * 0.01)
    eng = _engine_from_oracle(o, urm, B, hp)
    uids = rng.permutation(U)[:B]
    X = urm[uids].toarray()
    # D step
    ld_ref = o.d_step(uids, X)
    ld = eng.train_step(0, uids)
    assert o.hinge_active_last == (m > 1)
    assert abs(ld - ld_ref) <= 2e-5 * abs(ld_ref) + 1e-7
    for n in ("We", "be", "Wd", "bd"):
        _close(_get(eng, n), o.p[n], 2e-5, "D-step " + n)
    # G step on a different batch (ragged: B-3 rows)
    uids2 = rng.permutation(U)[:max(B - 3, 1)]
    X2 = urm[uids2].toarray()
    lg_ref = o.g_step(uids2, X2)
    lg = eng.train_step(1, uids2)
    assert abs(lg - lg_ref) <= 2e-5 * abs(lg_ref) + 1e-7
    for n in ("U", "V"):
        _close(_get(eng, n), o.p[n], 2e-5, "G-step " + n)
    np.testing.assert_allclose(eng.adam_powers(), [0.81, 0.998001, 0.81, 0.998001], rtol=1e-6)
    eng.close()


def test_adam_moments_after_steps():
    """m / v slots of every tensor after 3 D+G pairs (first-step Adam is sign-like, so the
    moments are the sharper check of gradient magnitudes)."""
    from ganmf_amd import _lib as L
    U, N, k, e, B = 120, 200, 16, 24, 32
    rng = np.random.RandomState(3)
    urm = _rand_urm(rng, U, N, 0.08)
    o = GANMFOracle(U, N, k, e, dtype=np.float64, seed=9, **HP)
    eng = _engine_from_oracle(o, urm, B, HP)
    for t in range(3):
        uids = rng.permutation(U)[:B]
        X = urm[uids].toarray()
        o.d_step(uids, X); eng.train_step(0, uids)
        o.g_step(uids, X); eng.train_step(1, uids)
    for n, opt in (("We", o.opt_d), ("be", o.opt_d), ("Wd", o.opt_d), ("bd", o.opt_d), ("U", o.opt_g), ("V", o.opt_g)):
        m_ref, v_ref = opt.slots[n]
        m = eng.get_tensor(NAME2ID[n], L.SLOT_ADAM_M).reshape(m_ref.shape)
        v = eng.get_tensor(NAME2ID[n], L.SLOT_ADAM_V).reshape(v_ref.shape)
        _close(m, m_ref, 5e-5, "adam m " + n)
        _close(v, v_ref, 5e-5, "adam v " + n)
    eng.close()


@pytest.mark.parametrize("d_steps,g_steps", [(1, 1), (2, 2)])
def test_epochs_match_oracle_ragged(d_steps, g_steps):
    """3 epochs of the reference schedule (cumulative in-place shuffles, ragged last batch)."""
    U, N, k, e, B = 101, 160, 12, 20, 16
    rng = np.random.RandomState(8)
    urm = _rand_urm(rng, U, N, 0.07)
    o = GANMFOracle(U, N, k, e, dtype=np.float64, seed=2, **HP)
    eng = _engine_from_oracle(o, urm, B, HP)
    for perm in reference_epoch_permutations(U, 3, 1337):
        dl_ref, gl_ref = o.train_epoch(urm, perm, B, d_steps, g_steps)
        dl, gl = eng.train_epoch(perm, d_steps, g_steps)
        assert len(dl) == d_steps * len(batch_slices(U, B)) and len(gl) == g_steps * len(batch_slices(U, B))
        np.testing.assert_allclose(dl, dl_ref, rtol=5e-5, atol=1e-7)
        np.testing.assert_allclose(gl, gl_ref, rtol=5e-5, atol=1e-7)
    for n in NAME2ID:
        _close(_get(eng, n), o.p[n], 1e-4, "epochs " + n)
    ids = np.arange(U)
    _close(eng.scores(ids), o.scores(ids), 1e-4, "scores")
    eng.close()


def test_golden_tiny_trajectory_user(golden_dir):
    """Committed golden vectors (oracle/make_golden.py): same init, same schedule."""
    g = np.load(os.path.join(golden_dir, "tiny_trajectories.npz"))
    urm = sps.load_npz(os.path.join(golden_dir, "tiny_urm.npz")).tocsr()
    name = "ganmf_user_hinge_on"
    c = json.loads(str(g[name + "/config"]))
    from ganmf_amd.engine import Engine
    eng = Engine(urm.shape[0], urm.shape[1], c["k"], c["e"], c["B"], **c["hp"])
    eng.set_urm(urm)
    for n, tid in NAME2ID.items():
        eng.set_tensor(tid, g["%s/init/%s" % (name, n)])
    dls, gls = [], []
    for perm in reference_epoch_permutations(urm.shape[0], c["epochs"], 1337):
        dl, gl = eng.train_epoch(perm, c["d_steps"], c["g_steps"])
        dls.append(dl); gls.append(gl)
    np.testing.assert_allclose(np.concatenate(dls), g[name + "/f64/dloss"], rtol=5e-5, atol=1e-7)
    np.testing.assert_allclose(np.concatenate(gls), g[name + "/f64/gloss"], rtol=5e-5, atol=1e-7)
    for n in NAME2ID:
        _close(_get(eng, n), g["%s/f64/final/%s" % (name, n)], 1e-4, "golden " + n)
    eng.close()


def test_item_mode_scores_and_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "tiny_trajectories.npz"))
    urm = sps.load_npz(os.path.join(golden_dir, "tiny_urm.npz")).tocsr()
    name = "ganmf_item_steps2"
    c = json.loads(str(g[name + "/config"]))
    from ganmf_amd.GANMF import GANMF
    np.random.seed(1337)
    model = GANMF(urm, mode='item', seed=11, is_experiment=True)
    model.initial_weights = {n: g["%s/init/%s" % (name, n)] for n in NAME2ID}
    ret = model.fit(num_factors=c["k"], emb_dim=c["e"], epochs=c["epochs"], batch_size=c["B"],
                    d_steps=c["d_steps"], g_steps=c["g_steps"], **c["hp"])
    assert ret == c["epochs"] + 1                       # return-value quirk (GANMF.py:244)
    assert model.URM_train.shape == urm.shape            # user x item on exit (GANMF.py:241-242)
    Uf, Vf = g[name + "/f64/final/U"], g[name + "/f64/final/V"]
    _close(model.user_factors(), Uf, 1e-4, "item-mode U")
    ids = np.array([0, 5, 36, 7])
    ref = (Uf @ Vf.T).T[ids]
    got = model._compute_item_score(ids)
    assert got.shape == (4, urm.shape[1]) and got.flags.writeable
    _close(got, ref, 1e-4, "item-mode scores")
    np.testing.assert_allclose(model.USER_factors[ids] @ model.ITEM_factors.T, got, rtol=1e-5, atol=1e-6)


def test_kat1_checkpoint_scores_and_ranking(golden_dir):
    """KAT-1: the reference's surviving checkpoint through the HIP scoring path + the build's
    recommend()/evaluator reproduce the reference's stored test_results.pkl."""
    from ganmf_amd.GANMF import GANMF
    from ganmf_amd.evaluation import EvaluatorHoldout
    t = np.load(os.path.join(golden_dir, "kat1_checkpoint_tensors.npz"))
    exp = json.load(open(os.path.join(golden_dir, "kat1_expected.json")))
    train = sps.load_npz(os.path.join(golden_dir, "LastFM_URM_train.npz")).tocsr()
    test = sps.load_npz(os.path.join(golden_dir, "LastFM_URM_test.npz")).tocsr()
    model = GANMF(train, mode='item', is_experiment=True)
    model._build(1, 133, 32)
    model.engine.set_tensor(100, t["U"])
    model.engine.set_tensor(101, t["V"])
    model.URM_train = model._URM_eval
    users = np.array(exp["users"])
    ranking = model.recommend(users, cutoff=50, remove_seen_flag=True)
    # k = 1: scores are exact products, rankings must agree wherever the reference scores differ
    same = sum(r == e for r, e in zip(ranking, exp["ranking_top50"]))
    assert same >= len(users) - 1, same
    res, _ = EvaluatorHoldout(test, [5, 10, 20, 50]).evaluateRecommender(model)
    for c, d in exp["expected_metrics"].items():
        for k in ("PRECISION", "RECALL", "MAP", "MRR", "NDCG", "F1", "HIT_RATE", "ARHR", "RMSE", "ROC_AUC",
                  "PRECISION_RECALL_MIN_DEN"):
            assert abs(res[int(c)][k] - d[k]) <= 1e-9 + 2e-6 * abs(d[k]), (c, k, res[int(c)][k], d[k])


def test_snapshot_restore_and_errors():
    from ganmf_amd import _lib as L
    from ganmf_amd.engine import Engine
    U, N = 50, 70
    rng = np.random.RandomState(1)
    urm = _rand_urm(rng, U, N, 0.1)
    o = GANMFOracle(U, N, 4, 6, seed=1, **HP)
    eng = _engine_from_oracle(o, urm, 16, HP)
    eng.snapshot_best()
    before = _get(eng, "V").copy()
    eng.train_step(1, np.arange(16))
    assert np.abs(_get(eng, "V") - before).max() > 0
    eng.restore_best()
    np.testing.assert_array_equal(_get(eng, "V"), before)
    with pytest.raises(L.GanmfError):
        eng.train_step(0, np.array([1, 1, 2]))          # duplicate ids
    with pytest.raises(L.GanmfError):
        eng.train_step(0, np.array([U + 3]))            # out of range
    with pytest.raises(L.GanmfError):
        eng.set_tensor(0, np.zeros(5, np.float32))      # wrong size
    with pytest.raises(L.GanmfError):
        eng.scores(np.array([-1]))
    eng.close()


@pytest.mark.parametrize("shape,lane_fence", [("small", None), ("c2", "0"), ("c2", "1")])
@pytest.mark.parametrize("force_collectives", [False, True])
def test_rccl_path_single_rank_matches_plain(monkeypatch, force_collectives, shape, lane_fence):
    """The data-parallel code path (RCCL all-reduce of the hinge sums, reduce-scatter of the D gradients and of gV, Adam on
    the rank's slice, all-gather of the parameters, presummed d_coef, per-epoch loss all-reduce) with a 1-rank communicator
    must reproduce the plain single-GPU path bit for bit.  force_collectives: GANMF_FORCE_COLLECTIVES=1 makes the one-rank
    communicator ISSUE its in-place ncclReduceScatter / ncclAllGather calls (they are skipped at world_size 1 otherwise),
    so the RCCL call sites of csrc/lib/dataparallel.inc reduce_scatter() / all_gather() execute on this one-GPU box.  shape "c2": the
    BASELINE configs[1] shape, where the data-parallel step takes the combined launches (de_dcoef_kernel, gWd + slab sum of
    dE, gUb + gV).  lane_fence: GANMF_LANE_EVENT_FENCE -- the lane events with (1: what world_size > 1 runs with) and without (0)
    the system-scope fence."""
    from ganmf_amd.engine import Engine, comm_unique_id
    if lane_fence is None:
        monkeypatch.delenv("GANMF_LANE_EVENT_FENCE", raising=False)
    else:
        monkeypatch.setenv("GANMF_LANE_EVENT_FENCE", lane_fence)
    if force_collectives:
        monkeypatch.setenv("GANMF_FORCE_COLLECTIVES", "1")
    else:
        monkeypatch.delenv("GANMF_FORCE_COLLECTIVES", raising=False)
    U, N, k, e, B = (150, 210, 9, 17, 32) if shape == "small" else (700, 3706, 250, 992, 128)
    rng = np.random.RandomState(5)
    urm = _rand_urm(rng, U, N, 0.08)
    o = GANMFOracle(U, N, k, e, seed=4, **HP)
    plain = _engine_from_oracle(o, urm, B, HP)
    dp = Engine(U, N, k, e, B, world_size=1, rank=0, **HP)
    dp.set_urm(urm)
    for n, tid in NAME2ID.items():
        dp.set_tensor(tid, o.p[n])
    dp.comm_init(comm_unique_id())
    perm = rng.permutation(U)
    steps = -(-U // B)
    rows = np.minimum(B, U - np.arange(steps) * B).astype(np.int32)
    for _ in range(2):
        dl0, gl0 = plain.train_epoch(perm, 1, 1)
        dl1, gl1 = dp.train_epoch(perm, 1, 1, steps_per_pass=steps, global_batch_rows=rows)
        np.testing.assert_array_equal(dl0, dl1)
        np.testing.assert_allclose(gl0, gl1, rtol=1e-6)
    for n in NAME2ID:
        np.testing.assert_array_equal(_get(plain, n), _get(dp, n))
    # a rank that ran out of rows: one extra step with zero local rows still opens the optimizer
    # step and joins every collective (what the other ranks' real rows would need)
    before = dp.adam_powers()
    dl2, gl2 = dp.train_epoch(perm, 1, 1, steps_per_pass=steps + 1, global_batch_rows=np.append(rows, 7).astype(np.int32))
    assert len(dl2) == steps + 1 and np.isfinite(dl2).all() and np.isfinite(gl2).all()
    np.testing.assert_allclose(dp.adam_powers()[0], before[0] * 0.9 ** (steps + 1), rtol=1e-5)
    plain.close(); dp.close()


def test_stream_timer_and_comm_info():
    """ganmf_stream_timer (bench.py's clock) and ganmf_comm_info (its `parallelism` object): the timer brackets what is enqueued
    between its two calls on the library's stream; comm_info reports (0, -1) without a communicator, RCCL's own (1, 0) for a
    one-rank communicator and the loopback group's size."""
    import time
    from ganmf_amd.engine import Engine, comm_unique_id
    rng = np.random.RandomState(3)
    U, N, k, e, B = 400, 600, 16, 40, 64
    urm = _rand_urm(rng, U, N, 0.05)
    eng = Engine(U, N, k, e, B, **HP)
    eng.set_urm(urm)
    assert eng.comm_info() == (0, -1)
    perm = rng.permutation(U)
    eng.train_epoch(perm, 1, 1)
    eng.timer_start()
    empty = eng.timer_stop()
    eng.timer_start()
    t0 = time.perf_counter()
    for _ in range(5):
        eng.train_epoch(perm, 1, 1)
    ms = eng.timer_stop()
    wall = (time.perf_counter() - t0) * 1e3
    assert 0.0 <= empty < 0.5 and ms > 5 * empty and 0.2 * wall < ms <= wall + 0.5, (empty, ms, wall)
    eng.close()
    dp = Engine(U, N, k, e, B, world_size=1, rank=0, **HP)
    dp.comm_init(comm_unique_id())
    assert dp.comm_info() == (1, 0)
    dp.close()
    a = Engine(U, N, k, e, B, world_size=2, rank=0, **HP)
    b = Engine(U, N, k, e, B, world_size=2, rank=1, **HP)
    a.comm_init_local(4242)
    b.comm_init_local(4242)
    assert a.comm_info() == (2, 0) and b.comm_info() == (2, 1)
    a.close(); b.close()


def test_rccl_comm_abort_stops_every_collective(monkeypatch):
    """ganmf_comm_abort on an RCCL communicator (ncclCommAbort frees it): from then on every entry that would touch the communicator --
    the collectives inside a training call (forced here on a one-rank communicator, so that the RCCL call sites execute) and
    ganmf_comm_info -- returns an error instead of handing freed memory to RCCL; abort is idempotent and the handle still closes."""
    from ganmf_amd._lib import GanmfError
    from ganmf_amd.engine import Engine, comm_unique_id
    monkeypatch.setenv("GANMF_FORCE_COLLECTIVES", "1")
    rng = np.random.RandomState(9)
    U, N, k, e, B = 150, 210, 9, 17, 32
    urm = _rand_urm(rng, U, N, 0.08)
    dp = Engine(U, N, k, e, B, world_size=1, rank=0, **HP)
    dp.set_urm(urm)
    dp.comm_init(comm_unique_id())
    perm = rng.permutation(U)
    steps = -(-U // B)
    rows = np.minimum(B, U - np.arange(steps) * B).astype(np.int32)
    dl, gl = dp.train_epoch(perm, 1, 1, steps_per_pass=steps, global_batch_rows=rows)
    assert np.isfinite(dl).all() and np.isfinite(gl).all()
    dp.comm_abort()
    dp.comm_abort()
    with pytest.raises(GanmfError, match="aborted"):
        dp.train_epoch(perm, 1, 1, steps_per_pass=steps, global_batch_rows=rows)
    with pytest.raises(GanmfError, match="aborted"):
        dp.comm_info()
    dp.close()
