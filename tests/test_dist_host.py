"""Host logic of the row-sharded fit() (ganmf_amd/dist.py) on CPU: the owner split of the reference's minibatch schedule,
and ShardedEngine's rank plumbing (in-process and one process per rank) over a recording stand-in engine.  The arithmetic
of the ranks is covered on the GPU (tests/test_gpu_sharded_fit.py, tests/test_gpu_dist_local.py) and, decomposed, on CPU
with gloo (tests/test_dist_gloo.py)."""
import numpy as np
import pytest
import scipy.sparse as sps

from ganmf_amd.dist import ShardedEngine, shard_bounds, split_by_owner
from tests.helpers_dist import T_ITEM_EMB, T_USER_EMB, failing_rank1_factory, recording_factory


@pytest.mark.parametrize("U,B,world", [(37, 8, 3), (64, 64, 4), (5, 8, 2), (100, 7, 8), (12, 5, 1)])
def test_split_by_owner_is_the_reference_schedule(U, B, world):
    rng = np.random.RandomState(U + B)
    perm = rng.permutation(U)
    bounds = shard_bounds(U, world)
    grows, per_rank = split_by_owner(perm, B, bounds)
    steps = -(-U // B)
    assert len(grows) == steps and grows.sum() == U
    assert all(g == min(B, U - i * B) for i, g in enumerate(grows))      # GANMF.py:177-189: ragged tail kept
    cursors = [0] * world
    for i in range(steps):
        ref = perm[i * B:(i + 1) * B]                                    # the reference's minibatch i
        got = []
        for r, (lo, hi) in enumerate(bounds):
            lp, lr = per_rank[r]
            assert len(lr) == steps and 0 <= lr[i] <= B
            mine = lp[cursors[r]:cursors[r] + lr[i]] + lo
            cursors[r] += lr[i]
            assert np.all((mine >= lo) & (mine < hi))
            # order of appearance inside the minibatch is kept
            assert list(mine) == [u for u in ref if lo <= u < hi]
            got.append(mine)
        assert sorted(np.concatenate(got)) == sorted(ref)                # union over ranks == the reference minibatch
        assert sum(per_rank[r][1][i] for r in range(world)) == grows[i]
    for r in range(world):
        assert cursors[r] == len(per_rank[r][0]) == bounds[r][1] - bounds[r][0]


def test_split_by_owner_allows_empty_steps_and_empty_epoch():
    bounds = shard_bounds(8, 2)
    grows, per_rank = split_by_owner(np.array([0, 1, 2, 3, 4, 5, 6, 7]), 4, bounds)     # unshuffled: each minibatch has ONE owner
    assert grows.tolist() == [4, 4]
    assert per_rank[0][1].tolist() == [4, 0] and per_rank[1][1].tolist() == [0, 4]
    grows, per_rank = split_by_owner(np.array([], dtype=np.int64), 4, bounds)
    assert len(grows) == 0 and all(len(lp) == 0 and len(lr) == 0 for lp, lr in per_rank)


def _drive(backend, world, factory=recording_factory):
    U, N, k, e, B = 23, 11, 3, 4, 5
    devices = list(range(world)) if backend == "process" else [0]
    eng = ShardedEngine(U, N, k, e, B, world_size=None if backend == "process" else world, devices=devices,
                        backend=backend, engine_factory=factory, d_lr=1e-3)
    urm = sps.random(U, N, density=0.3, format="csr", random_state=1, dtype=np.float32)
    eng.set_urm(urm)
    eng.set_seen(urm)
    u0 = np.arange(U * k, dtype=np.float32).reshape(U, k) / 1000.0
    eng.set_tensor(T_USER_EMB, u0)
    eng.set_tensor(T_ITEM_EMB, np.ones((N, k), np.float32))
    return eng, u0, (U, N, k, e, B)


@pytest.mark.parametrize("backend,world", [("local", 3), ("process", 2)])
def test_sharded_engine_plumbing(backend, world):
    eng, u0, (U, N, k, e, B) = _drive(backend, world)
    try:
        assert len(eng.ranks) == world and eng.bounds == shard_bounds(U, world)
        perm = np.random.RandomState(0).permutation(U)
        dl, gl = eng.train_epoch(perm, 1, 1)
        steps = -(-U // B)
        assert len(dl) == steps and len(gl) == steps
        # the rank-owned rows come back in global order: every row visited exactly once -> +1 everywhere
        got = eng.get_tensor(T_USER_EMB)
        np.testing.assert_allclose(got, u0 + 1.0, rtol=1e-6)
        np.testing.assert_allclose(eng.get_tensor(T_ITEM_EMB), 1.0 + steps)       # replicated: rank 0's copy
        s = eng.scores(np.array([0, U - 1]))
        np.testing.assert_allclose(s, (u0[[0, U - 1]] + 1.0) @ np.full((N, k), 1.0 + steps, np.float32).T, rtol=1e-6)
        # snapshot -> train -> restore: the best parameters go back to every rank
        eng.snapshot_best()
        eng.train_epoch(np.random.RandomState(1).permutation(U), 1, 1)
        np.testing.assert_allclose(eng.get_tensor(T_USER_EMB), u0 + 2.0, rtol=1e-6)
        eng.restore_best()
        np.testing.assert_allclose(eng.get_tensor(T_USER_EMB), u0 + 1.0, rtol=1e-6)
        eng.train_epoch(np.random.RandomState(2).permutation(U), 1, 1)             # the ranks continue from the restored state
        np.testing.assert_allclose(eng.get_tensor(T_USER_EMB), u0 + 2.0, rtol=1e-6)
        np.testing.assert_allclose(eng.get_tensor(T_ITEM_EMB), 1.0 + 2 * steps)
    finally:
        eng.close()
    assert eng.ranks == [] and eng.master is None


def test_sharded_engine_rank_failure_ends_the_group():
    eng, _, (U, *_rest) = _drive("process", 2, factory=failing_rank1_factory)
    from ganmf_amd._lib import GanmfError
    with pytest.raises(GanmfError, match="synthetic failure on rank 1"):
        eng.train_epoch(np.arange(U), 1, 1)
    assert eng.ranks == [] and eng.master is None      # closed: no rank process is left waiting in a collective


def test_sharded_engine_argument_checks():
    with pytest.raises(ValueError):
        ShardedEngine(10, 5, 2, 2, 4, devices=[0, 0], backend="process", engine_factory=recording_factory)
    with pytest.raises(ValueError):
        ShardedEngine(3, 5, 2, 2, 4, world_size=4, backend="local", engine_factory=recording_factory)
    with pytest.raises(ValueError):
        ShardedEngine(10, 5, 2, 2, 4, backend="threads", engine_factory=recording_factory)


def test_host_class_reads_the_sharding_request_from_the_environment(monkeypatch):
    """The reference's drivers construct GANMF(URM_train, mode=..., seed=..., is_experiment=True) (RunBestParameters.py:86-88): the
    row-sharded fit is asked for through GANMF_DEVICES / GANMF_DIST_BACKEND / GANMF_WORLD_SIZE without touching them; explicit
    constructor arguments win."""
    import scipy.sparse as sps
    from ganmf_amd.DisGANMF import DisGANMF
    from ganmf_amd.GANMF import GANMF
    urm = sps.identity(12, format="csr", dtype=np.float32)
    for k in ("GANMF_DEVICES", "GANMF_DIST_BACKEND", "GANMF_WORLD_SIZE"):
        monkeypatch.delenv(k, raising=False)
    m = GANMF(urm, mode="user", is_experiment=True)
    assert m.devices is None and m.dist_backend == "process" and m.world_size is None and not m._sharded()
    monkeypatch.setenv("GANMF_DEVICES", "0, 1,2,3")
    for cls in (GANMF, DisGANMF):
        m = cls(urm, mode="item", is_experiment=True)
        assert m.devices == [0, 1, 2, 3] and m._sharded()
    assert GANMF(urm, is_experiment=True, devices=[5]).devices == [5] and not GANMF(urm, is_experiment=True, devices=[5])._sharded()
    monkeypatch.delenv("GANMF_DEVICES")
    monkeypatch.setenv("GANMF_DIST_BACKEND", "local")
    monkeypatch.setenv("GANMF_WORLD_SIZE", "4")
    m = GANMF(urm, is_experiment=True)
    assert m.dist_backend == "local" and m.world_size == 4 and m._sharded()
    assert not GANMF(urm, is_experiment=True, world_size=1)._sharded()
    assert GANMF(urm, is_experiment=True, dist_backend="process").dist_backend == "process"
