"""The data-parallel path of the library with world_size > 1 on ONE GPU, through the in-process loopback
communicator (ganmf_comm_init_local): one host thread per rank, each with its own engine, shard of the CSR matrix,
rows of U and row_offset.  One epoch (a D pass then a G pass over ragged, unequal shards — one rank runs out of rows
before the others) must reproduce the single-process oracle on the UNION batches: replicated tensors identical on
every rank (bitwise) and equal to the oracle's, every rank's rows of U equal to the oracle's rows, and the losses
every rank reports equal to the union-batch losses.  The RCCL communicator takes the same code path with
ncclAllReduce in place of the rendezvous (tests/test_gpu_parity.py runs it with one rank)."""
import threading

import numpy as np
import pytest
import scipy.sparse as sps

from ganmf_amd.dist import epoch_plan, shard_bounds
from oracle.ganmf_oracle import DisGANMFOracle, GANMFOracle

pytestmark = pytest.mark.gpu


def _err(got, ref):
    return np.max(np.abs(np.asarray(got, np.float64).reshape(np.shape(ref)) - ref)) / (np.max(np.abs(ref)) + 1e-30)


def _run_ranks(world, make_engine, bounds, perms, B, group):
    from ganmf_amd import _lib as L
    steps, grows = epoch_plan([b - a for a, b in bounds], B)
    out, errs = [None] * world, []
    engines = [make_engine(r) for r in range(world)]
    for e in engines:
        e.comm_init_local(group)

    def work(r):
        try:
            out[r] = engines[r].train_epoch(perms[r], 1, 1, steps_per_pass=steps, global_batch_rows=grows)
        except Exception as ex:      # a failed rank must not leave its peers in the rendezvous
            errs.append((r, ex))
    threads = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(300)
    assert not errs, errs
    assert all(not t.is_alive() for t in threads)
    return engines, out, steps


@pytest.mark.parametrize("world,U", [(2, 37), (3, 50)])
def test_ganmf_sharded_epoch_equals_union_batch(world, U):
    from ganmf_amd.engine import Engine
    N, k, e, B = 61, 5, 9, 8
    rng = np.random.RandomState(world)
    urm = sps.csr_matrix((rng.rand(U, N) < 0.15).astype(np.float32))
    hp = dict(d_lr=1e-3, g_lr=2e-3, d_reg=1e-3, g_reg=1e-4, m=10.0, recon_coefficient=0.2)
    o = GANMFOracle(U, N, k, e, dtype=np.float64, seed=3, **hp)
    o.set_params(be=rng.randn(e) * 0.01, bd=rng.randn(N) * 0.01)
    p0 = o.get_params()
    bounds = shard_bounds(U, world)
    perms = [rng.permutation(b - a) for a, b in bounds]

    def make_engine(r):
        lo, hi = bounds[r]
        eng = Engine(hi - lo, N, k, e, B, world_size=world, rank=r, row_offset=lo, **hp)
        eng.set_urm(urm[lo:hi])
        for n, tid in (("We", 0), ("be", 1), ("Wd", 2), ("bd", 3), ("V", 101)):
            eng.set_tensor(tid, p0[n])
        eng.set_tensor(100, p0["U"][lo:hi])
        return eng

    engines, out, steps = _run_ranks(world, make_engine, bounds, perms, B, group=100 + world)
    # oracle on the union batches: D pass, then G pass, over the same slices
    unions = [np.concatenate([bounds[r][0] + perms[r][i * B:(i + 1) * B] for r in range(world)]) for i in range(steps)]
    dl_ref = [o.d_step(u, urm[u].toarray()) for u in unions]
    gl_ref = [o.g_step(u, urm[u].toarray()) for u in unions]
    for r in range(world):
        np.testing.assert_allclose(out[r][0], dl_ref, rtol=1e-4, atol=1e-7)
        np.testing.assert_allclose(out[r][1], gl_ref, rtol=1e-4, atol=1e-7)
    for n, tid in (("We", 0), ("be", 1), ("Wd", 2), ("bd", 3), ("V", 101)):
        t0 = engines[0].get_tensor(tid)
        assert _err(t0, o.p[n]) <= 1e-4, n
        for r in range(1, world):
            assert np.array_equal(engines[r].get_tensor(tid), t0), (n, r)       # replicas stay identical
    for r, (lo, hi) in enumerate(bounds):
        assert _err(engines[r].get_tensor(100), o.p["U"][lo:hi]) <= 1e-4, r
    for eng in engines:
        eng.close()


def test_disganmf_sharded_epoch_equals_union_batch():
    from ganmf_amd import _lib as L
    from ganmf_amd.engine import Engine
    world, U, N, k, e, B = 2, 29, 40, 4, 7, 8
    rng = np.random.RandomState(9)
    urm = sps.csr_matrix((rng.rand(U, N) < 0.2).astype(np.float32))
    hp = dict(d_lr=1e-3, g_lr=1e-3, d_reg=1e-4, g_reg=0.0, recon_coefficient=0.3)
    o = DisGANMFOracle(U, N, k, d_layers=2, d_nodes=e, d_hidden_act="tanh", dtype=np.float64, seed=2, **hp)
    p0 = {n: v.copy() for n, v in o.p.items()}
    bounds = shard_bounds(U, world)
    perms = [rng.permutation(b - a) for a, b in bounds]
    ids = {"W0": 0, "b0": 1, "W1": 2, "b1": 3, "Wo": 4, "bo": 5, "V": 101}

    def make_engine(r):
        lo, hi = bounds[r]
        eng = Engine(hi - lo, N, k, e, B, model=L.MODEL_DISGANMF, d_layers=2, d_act="tanh", m=0.0, world_size=world, rank=r,
                     row_offset=lo, **hp)     # row_offset: the net is fed float(GLOBAL uid) (DisGANMF.py:110)
        eng.set_urm(urm[lo:hi])
        for n, tid in ids.items():
            eng.set_tensor(tid, p0[n])
        eng.set_tensor(100, p0["U"][lo:hi])
        return eng

    engines, out, steps = _run_ranks(world, make_engine, bounds, perms, B, group=77)
    unions = [np.concatenate([bounds[r][0] + perms[r][i * B:(i + 1) * B] for r in range(world)]) for i in range(steps)]
    dl_ref = [o.d_step(u, urm[u].toarray()) for u in unions]
    gl_ref = [o.g_step(u, urm[u].toarray()) for u in unions]
    for r in range(world):
        np.testing.assert_allclose(out[r][0], dl_ref, rtol=5e-4, atol=1e-6)
        np.testing.assert_allclose(out[r][1], gl_ref, rtol=5e-4, atol=1e-6)
    for n, tid in ids.items():
        t0 = engines[0].get_tensor(tid)
        assert _err(t0, o.p[n]) <= 5e-4, n
        assert np.array_equal(engines[1].get_tensor(tid), t0), n
    for r, (lo, hi) in enumerate(bounds):
        assert _err(engines[r].get_tensor(100), o.p["U"][lo:hi]) <= 5e-4, r
    for eng in engines:
        eng.close()


@pytest.mark.slow
def test_c4_width_world8_step_equals_union_batch():
    """BASELINE configs[3] geometry as a MULTI-RANK run: 50 000 items, k = 250, emb_dim = 1024, B = 128 per rank, users
    sharded over 8 ranks (2000 users -> 250 rows per rank: a full slice of 128 and a ragged one of 122 per rank, global
    batches of 1024 and 976 rows, every scale uses them), a discriminator pass and a generator pass of two updates each through
    the library's data-parallel path on the loopback communicator (world 8 on one GPU), against the single-process oracle on the
    union batches.  Replicated tensors must be bitwise identical on all 8 ranks."""
    from ganmf_amd.engine import Engine
    world, U, N, k, e, B = 8, 2000, 50000, 250, 1024, 128
    rng = np.random.RandomState(4)
    nnz_per_row = 500                                   # 1 % density (C4)
    cols = np.concatenate([rng.choice(N, nnz_per_row, replace=False) for _ in range(U)])
    rows = np.repeat(np.arange(U), nnz_per_row)
    urm = sps.csr_matrix((np.ones(len(cols), np.float32), (rows, cols)), shape=(U, N))
    hp = dict(d_lr=1e-4, g_lr=1e-3, d_reg=1e-5, g_reg=0.0, m=10.0, recon_coefficient=0.2)
    o = GANMFOracle(U, N, k, e, dtype=np.float64, seed=5, **hp)
    p0 = o.get_params()
    bounds = shard_bounds(U, world)
    perms = [rng.permutation(b - a) for a, b in bounds]

    def make_engine(r):
        lo, hi = bounds[r]
        eng = Engine(hi - lo, N, k, e, B, world_size=world, rank=r, row_offset=lo, **hp)
        eng.set_urm(urm[lo:hi])
        for n, tid in (("We", 0), ("be", 1), ("Wd", 2), ("bd", 3), ("V", 101)):
            eng.set_tensor(tid, p0[n])
        eng.set_tensor(100, p0["U"][lo:hi])
        return eng

    engines, out, steps = _run_ranks(world, make_engine, bounds, perms, B, group=808)
    assert steps == 2
    unions = [np.concatenate([bounds[r][0] + perms[r][i * B:(i + 1) * B] for r in range(world)]) for i in range(steps)]
    assert [len(u) for u in unions] == [1024, 976]
    dl_ref = [o.d_step(u, urm[u].toarray()) for u in unions]
    gl_ref = [o.g_step(u, urm[u].toarray()) for u in unions]
    for r in range(world):
        np.testing.assert_allclose(out[r][0], dl_ref, rtol=1e-4, atol=1e-7)
        np.testing.assert_allclose(out[r][1], gl_ref, rtol=1e-4, atol=1e-7)
    for n, tid in (("We", 0), ("be", 1), ("Wd", 2), ("bd", 3), ("V", 101)):
        t0 = engines[0].get_tensor(tid)
        # the first Adam steps move every element by ~lr whatever the gradient: compare the UPDATE, normalised by its own scale
        upd, upd_ref = t0.astype(np.float64).reshape(p0[n].shape) - p0[n], o.p[n] - p0[n]
        assert np.max(np.abs(upd - upd_ref)) <= 4e-3 * np.max(np.abs(upd_ref)) + 1e-12, n
        assert _err(t0, o.p[n]) <= 1e-4, n
        for r in range(1, world):
            assert np.array_equal(engines[r].get_tensor(tid), t0), (n, r)
    for r, (lo, hi) in enumerate(bounds):
        assert _err(engines[r].get_tensor(100), o.p["U"][lo:hi]) <= 1e-4, r
    for eng in engines:
        eng.close()


def test_local_group_errors():
    from ganmf_amd import _lib as L
    from ganmf_amd.engine import Engine
    a = Engine(8, 9, 2, 3, 4, world_size=2, rank=0)
    a.comm_init_local(5)
    with pytest.raises(L.GanmfError, match="already has a communicator"):
        a.comm_init_local(5)
    b = Engine(8, 9, 2, 3, 4, world_size=3, rank=1)
    with pytest.raises(L.GanmfError, match="world_size"):
        b.comm_init_local(5)
    a.close(); b.close()


def test_comm_abort_releases_a_rank_waiting_for_a_failed_peer():
    """ganmf_comm_abort (the one entry point that may be called from another thread than the one inside a training call): rank 0 of a
    two-rank loopback group enters its epoch, rank 1 never does (it "failed" on the host side); the driver aborts rank 0's communicator
    and the training call returns with an error instead of waiting out the rendezvous -- and only THEN are the engines destroyed
    (ganmf_amd/dist.py _ThreadRank.kill: never free a handle under a thread that is still inside the library)."""
    import time
    from ganmf_amd import _lib as L
    from ganmf_amd.dist import _ThreadRank
    from ganmf_amd.engine import Engine
    U, N, k, e, B = 40, 61, 5, 9, 8
    rng = np.random.RandomState(0)
    urm = sps.csr_matrix((rng.rand(U, N) < 0.2).astype(np.float32))
    hp = dict(d_lr=1e-3, g_lr=1e-3, d_reg=0.0, g_reg=0.0, m=1.0, recon_coefficient=0.1)
    bounds = shard_bounds(U, 2)

    def factory(rank):
        lo, hi = bounds[rank]
        eng = Engine(hi - lo, N, k, e, B, world_size=2, rank=rank, row_offset=lo, **hp)
        eng.set_urm(urm[lo:hi])
        return eng

    ranks = [_ThreadRank(lambda r=r, **kw: factory(r)) for r in range(2)]
    for rk in ranks:
        rk.submit("__create__")
    for rk in ranks:
        rk.result(60)
    for rk in ranks:
        rk.submit("comm_init_local", 4242)
    for rk in ranks:
        rk.result(60)
    steps, grows = epoch_plan([b - a for a, b in bounds], B)
    ranks[0].submit("train_epoch", rng.permutation(bounds[0][1] - bounds[0][0]), 1, 1, steps_per_pass=steps, global_batch_rows=grows)
    time.sleep(1.0)                      # rank 0 is inside the library now, waiting for rank 1 in the first collective
    assert not ranks[0].ready(0.0)
    t0 = time.time()
    ranks[0].kill()                      # comm_abort -> the call returns -> the engine is closed
    assert time.time() - t0 < 30
    assert ranks[0].eng is None and not ranks[0]._t.is_alive()
    status, err = ranks[0]._out
    assert status == "err" and isinstance(err, L.GanmfError) and "peer failed" in str(err)
    ranks[1].kill()
    assert ranks[1].eng is None
