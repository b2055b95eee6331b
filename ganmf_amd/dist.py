"""Row-wise (user) sharding for data-parallel training on the GPUs of one node.

Every rank owns a contiguous block of the generator's users: its CSR rows, its rows of
user_embeddings and their Adam moments (never communicated).  item_embeddings and all
discriminator tensors are replicated; inside libganmf_hip their gradients are reduce-scattered
(RCCL over xGMI) every step, each rank applies TF-Adam to its slice -- the moments of a replicated
tensor therefore live SHARDED across the ranks, ganmf_get_tensor rejects those slots -- and the
updated parameters are all-gathered, so the replicas stay bitwise identical.

Step agreement: every rank must issue the same collectives, so all ranks run the same number of
steps per pass; a rank without rows in a step contributes zero gradients.  All losses are *global*
means, so each step needs the global number of rows of its minibatch -- computable by every rank
from the schedule alone (no communication).

Two schedules:
  * `epoch_plan` / `local_permutation` -- weak scaling (bench.py): every rank shuffles its own rows and cuts
    slices of batch_size, global minibatch = world x batch_size;
  * `split_by_owner` -- what `ShardedEngine.train_epoch` (the sharded fit()) uses: the REFERENCE's schedule,
    one shuffled permutation of all rows cut into minibatches of batch_size (GANMF.py:175-203), each
    minibatch split by row owner.  The union over ranks of step i is exactly the reference's minibatch i, so a
    sharded fit() follows the single-GPU trajectory up to fp32 summation order, whatever the world size.

`ShardedEngine` stands where the host classes keep their `Engine` (same methods): `world_size` rank engines --
one process per GPU over an RCCL communicator, or (tests, one GPU) in-process engines on the loopback
communicator -- plus a full-size "master" engine on the first device that receives the rank-owned rows of
user_embeddings and one copy of the replicated tensors whenever scores, recommendations, evaluation, a snapshot
or persistence are asked for (GANMF.py:285-292: `_compute_item_score` needs all of both factor matrices).
"""
import multiprocessing as mp
import threading
import time

import numpy as np

from . import _lib as L


def shard_bounds(n_rows, world_size):
    """Contiguous, balanced: the first (n_rows % world) ranks get one extra row."""
    base, extra = divmod(n_rows, world_size)
    sizes = np.array([base + (1 if r < extra else 0) for r in range(world_size)], dtype=np.int64)
    starts = np.concatenate([[0], np.cumsum(sizes)[:-1]])
    return [(int(s), int(s + n)) for s, n in zip(starts, sizes)]


def epoch_plan(shard_sizes, batch_size):
    """steps per pass and, per step, the number of rows summed over ranks."""
    shard_sizes = np.asarray(shard_sizes, dtype=np.int64)
    steps = int(np.max(-(-shard_sizes // batch_size)))
    i = np.arange(steps)[:, None]
    rows = np.clip(shard_sizes[None, :] - i * batch_size, 0, batch_size)
    return steps, rows.sum(axis=1).astype(np.int32)


def local_permutation(n_local, seed, rank, epoch_state=None):
    """Per-rank shuffle stream RandomState(seed + rank); in place and cumulative across epochs
    like the reference's single stream (GANMF.py:156,175)."""
    if epoch_state is None:
        epoch_state = (np.random.RandomState(seed + rank), np.arange(n_local))
    rng, perm = epoch_state
    rng.shuffle(perm)
    return perm.copy(), epoch_state


def split_by_owner(perm, batch_size, bounds):
    """The reference's epoch schedule on row-sharded ranks.

    `perm`: this epoch's shuffled GLOBAL row ids (GANMF.py:175); minibatch i = perm[i*B:(i+1)*B] (ragged tail kept,
    :177-203).  Returns (global_rows, per_rank) where global_rows[i] = rows of minibatch i and per_rank[r] =
    (local_perm, local_rows): the LOCAL ids (global id - bounds[r][0]) of the rows rank r owns, minibatch by minibatch in
    their order of appearance, and how many of them belong to each minibatch (0 is possible)."""
    perm = np.asarray(perm, dtype=np.int64)
    n, B = perm.size, int(batch_size)
    steps = -(-n // B) if n else 0
    step_of = np.arange(n) // B
    global_rows = np.bincount(step_of, minlength=steps).astype(np.int32)
    starts = np.array([b[0] for b in bounds], dtype=np.int64)
    owner = np.searchsorted(starts, perm, side="right") - 1
    per_rank = []
    for r, (lo, hi) in enumerate(bounds):
        mine = owner == r              # boolean selection keeps the order of appearance
        local = (perm[mine] - lo).astype(np.int32)
        assert local.size == 0 or (local.min() >= 0 and local.max() < hi - lo)
        per_rank.append((local, np.bincount(step_of[mine], minlength=steps).astype(np.int32)))
    return global_rows, per_rank


# ---- rank back-ends ---------------------------------------------------------------------------------
def _default_engine_factory(**kw):
    from .engine import Engine
    return Engine(**kw)


def _rank_process(conn, factory):
    """Body of one rank process: owns one engine, executes (method, args, kwargs) requests from the driver."""
    eng = None
    try:
        while True:
            msg = conn.recv()
            if msg is None:
                break
            name, args, kw = msg
            try:
                if name == "__create__":
                    eng = factory(**kw)
                    out = None
                elif name == "__unique_id__":
                    out = eng.comm_unique_id()
                else:
                    out = getattr(eng, name)(*args, **kw)
                conn.send(("ok", out))
            except BaseException as ex:      # reported to the driver, which ends the whole group
                conn.send(("err", "%s: %s" % (type(ex).__name__, ex)))
    finally:
        if eng is not None:
            try:
                eng.close()
            except Exception:
                pass
        conn.close()


class _ProcessRank(object):
    """One rank = one spawned process (never forked: the driver may have touched the GPU) bound to one GPU."""

    def __init__(self, factory):
        ctx = mp.get_context("spawn")
        self.conn, child = ctx.Pipe()
        self.proc = ctx.Process(target=_rank_process, args=(child, factory), daemon=True)
        self.proc.start()
        child.close()

    def submit(self, name, *args, **kw):
        self.conn.send((name, args, kw))

    def ready(self, timeout):
        """an answer (or the end of the pipe: the process died) is waiting"""
        return self.conn.poll(timeout)

    def kill(self):
        """end the process now (a peer failed: this rank may sit in a collective that will never complete)"""
        try:
            self.conn.close()
        except Exception:
            pass
        if self.proc.is_alive():
            self.proc.terminate()
            self.proc.join(timeout=10)
            if self.proc.is_alive():
                self.proc.kill()

    def result(self, timeout=600.0):
        if not self.conn.poll(timeout):
            raise L.GanmfError("sharded fit: a rank process did not answer within %.0f s" % timeout)
        try:
            status, out = self.conn.recv()
        except EOFError:
            raise L.GanmfError("sharded fit: a rank process died (exit code %s)" % self.proc.exitcode)
        if status != "ok":
            if "MemoryError" in out:
                raise MemoryError(out)
            raise L.GanmfError("sharded fit: rank failed: " + out)
        return out

    def close(self):
        try:
            self.conn.send(None)
        except Exception:
            pass
        self.proc.join(timeout=30)
        if self.proc.is_alive():
            self.proc.terminate()


class _ThreadRank(object):
    """One rank = one engine of THIS process, driven by its own host thread per request (the loopback communicator's
    collectives are rendezvous between those threads; ctypes releases the GIL inside the library)."""

    def __init__(self, factory):
        self.factory, self.eng, self._t, self._out = factory, None, None, None

    def submit(self, name, *args, **kw):
        def run():
            try:
                if name == "__create__":
                    self.eng = self.factory(**kw)
                    self._out = ("ok", None)
                else:
                    self._out = ("ok", getattr(self.eng, name)(*args, **kw))
            except BaseException as ex:
                self._out = ("err", ex)
        self._t = threading.Thread(target=run)
        self._t.start()

    def ready(self, timeout):
        self._t.join(timeout)
        return not self._t.is_alive()

    def kill(self):
        """A peer failed.  Threads cannot be ended, and the engine must not be destroyed under a thread that is still inside a
        library call on it: the loopback group is marked failed (ganmf_comm_abort touches nothing but the group's own state),
        which makes the call in flight return with an error; the engine is closed once its thread has come back."""
        eng = self.eng
        if eng is not None and hasattr(eng, "comm_abort"):
            try:
                eng.comm_abort()
            except Exception:
                pass
        if self._t is not None:
            self._t.join(150.0)      # (the loopback rendezvous itself gives up after 120 s)
            if self._t.is_alive():
                return               # never free a handle that is in use: leak it instead
        self.close()

    def result(self, timeout=600.0):
        self._t.join(timeout)
        if self._t.is_alive():
            raise L.GanmfError("sharded fit: a rank thread did not finish within %.0f s" % timeout)
        status, out = self._out
        if status != "ok":
            raise out
        return out

    def close(self):
        if self._t is not None and self._t.is_alive():      # a request still inside the library (kill() gave up on it): the handle stays
            return
        if self.eng is not None:
            self.eng.close()
            self.eng = None


_group_ids = iter(range(7000, 1 << 30))      # loopback communicator ids of this process


class ShardedEngine(object):
    """`world_size` row-sharded rank engines + one master engine; the methods the host classes call on an Engine.

    devices:  list of HIP device ids, one rank PROCESS per entry over RCCL (`backend="process"`, the production form), or
    backend="local": `world_size` in-process engines on `devices[0]` over the loopback communicator (one GPU; tests).
    engine_factory(**engine_kwargs) -> engine: injectable (tests); the default builds ganmf_amd.engine.Engine and
    therefore fails loudly without the HIP library and a GPU."""

    def __init__(self, num_users, num_items, num_factors, emb_dim, batch_size, world_size=None, devices=None,
                 backend="process", engine_factory=None, **engine_kw):
        devices = list(devices) if devices is not None else [0]
        if backend not in ("process", "local"):
            raise ValueError("ShardedEngine: backend must be 'process' or 'local'")
        if backend == "process":
            if world_size is not None and world_size != len(devices):
                raise ValueError("ShardedEngine: one rank process per device: world_size %r != %d devices" % (world_size, len(devices)))
            world_size = len(devices)
            if len(set(devices)) != len(devices):
                raise ValueError("ShardedEngine: RCCL takes one rank per GPU; use backend='local' to put several ranks on one")
        world_size = int(world_size or 1)
        if not (1 <= world_size <= num_users):
            raise ValueError("ShardedEngine: world_size %d for %d rows" % (world_size, num_users))
        self.world, self.backend, self.devices = world_size, backend, devices
        self.num_users, self.num_items = num_users, num_items
        self.batch_size = min(batch_size, num_users)
        self.bounds = shard_bounds(num_users, world_size)
        factory = engine_factory or _default_engine_factory
        self._dirty = False
        self.master = None
        self.ranks = []
        try:
            # the master first: sized for ALL users; scoring, recommend, evaluation, snapshots and persistence run on it
            self.master = factory(num_users=num_users, num_items=num_items, num_factors=num_factors, emb_dim=emb_dim,
                                  batch_size=batch_size, device=devices[0], **engine_kw)
            Rank = _ProcessRank if backend == "process" else _ThreadRank
            self.ranks = [Rank(factory) for _ in range(world_size)]
            for r, rk in enumerate(self.ranks):
                lo, hi = self.bounds[r]
                rk.submit("__create__", num_users=hi - lo, num_items=num_items, num_factors=num_factors, emb_dim=emb_dim,
                          batch_size=batch_size, device=devices[r] if backend == "process" else devices[0],
                          world_size=world_size, rank=r, row_offset=lo, **engine_kw)
            self._collect()
            if backend == "process":
                self.ranks[0].submit("__unique_id__")
                uid = self.ranks[0].result()
                self._all("comm_init", uid)
            else:
                self._all("comm_init_local", next(_group_ids))
        except BaseException:
            self.close()
            raise

    # ---- plumbing -------------------------------------------------------------------------------------
    def _collect(self, timeout=600.0):
        """results of the request every rank was just handed.  All ranks are polled together; the FIRST failure (an error
        reply, a dead process, the deadline) ends the whole group at once -- the survivors sit in a collective that can no
        longer complete, and waiting for them one after the other would stall the caller for world x timeout."""
        n = len(self.ranks)
        outs, first, pending = [None] * n, None, list(range(n))
        deadline = time.monotonic() + timeout
        while pending and first is None:
            for i in list(pending):
                if self.ranks[i].ready(0.02):
                    try:
                        outs[i] = self.ranks[i].result(timeout=5.0)
                    except BaseException as ex:
                        first = ex
                        break
                    pending.remove(i)
            if pending and first is None and time.monotonic() > deadline:
                first = L.GanmfError("sharded fit: rank(s) %s did not answer within %.0f s" % (pending, timeout))
        if first is not None:
            for rk in self.ranks:
                try:
                    rk.kill()
                except Exception:
                    pass
            self.close()
            raise first
        return outs

    def _all(self, name, *args, **kw):
        for rk in self.ranks:
            rk.submit(name, *args, **kw)
        return self._collect()

    def _each(self, name, per_rank_args):
        for rk, a in zip(self.ranks, per_rank_args):
            rk.submit(name, *a)
        return self._collect()

    def close(self):
        for rk in self.ranks:
            try:
                rk.close()
            except Exception:
                pass
        self.ranks = []
        if self.master is not None:
            self.master.close()
            self.master = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- data and parameters --------------------------------------------------------------------------
    def set_urm(self, urm_csr):
        urm = urm_csr.tocsr()
        if urm.shape != (self.num_users, self.num_items):
            raise ValueError("ShardedEngine.set_urm: shape %r" % (urm.shape,))
        # (the master never trains: it gets no copy of the matrix)
        self._each("set_urm", [(urm[lo:hi],) for lo, hi in self.bounds])

    def shape(self, tid):
        return self.master.shape(tid)

    def set_tensor(self, tid, arr, slot=L.SLOT_PARAM):
        a = np.ascontiguousarray(arr, dtype=np.float32)
        self.master.set_tensor(tid, a, slot)
        if tid == L.T_USER_EMB:
            a2 = a.reshape(self.num_users, -1)
            self._each("set_tensor", [(tid, a2[lo:hi], slot) for lo, hi in self.bounds])
        else:
            if slot in (L.SLOT_ADAM_M, L.SLOT_ADAM_V) and self.world > 1:
                raise L.GanmfError("ShardedEngine: the Adam moments of a replicated tensor are sharded over the ranks")
            self._all("set_tensor", tid, a, slot)

    def _sync_master(self):
        """rank-owned rows of user_embeddings + one copy of every replicated tensor -> the master engine"""
        if not self._dirty:
            return
        parts = self._all("get_tensor", L.T_USER_EMB)
        self.master.set_tensor(L.T_USER_EMB, np.concatenate(parts, axis=0))
        tid = 0
        tids = [L.T_ITEM_EMB]
        while True:                       # discriminator tensors are numbered 0 .. until the library says "unknown"
            try:
                self.master.shape(tid)
            except L.GanmfError:
                break
            tids.append(tid)
            tid += 1
        for t in tids:
            self.ranks[0].submit("get_tensor", t)
            self.master.set_tensor(t, self.ranks[0].result())
        self._dirty = False

    def get_tensor(self, tid, slot=L.SLOT_PARAM):
        if slot in (L.SLOT_ADAM_M, L.SLOT_ADAM_V):
            if tid != L.T_USER_EMB and self.world > 1:
                raise L.GanmfError("ShardedEngine: the Adam moments of a replicated tensor are sharded over the ranks")
            src = self._all("get_tensor", tid, slot) if tid == L.T_USER_EMB else [self._all("get_tensor", tid, slot)[0]]
            return np.concatenate(src, axis=0)
        self._sync_master()
        return self.master.get_tensor(tid, slot)

    def adam_powers(self):
        self.ranks[0].submit("adam_powers")
        return self.ranks[0].result()

    # ---- training ---------------------------------------------------------------------------------------
    def train_epoch(self, perm, d_steps=1, g_steps=1, steps_per_pass=0, global_batch_rows=None):
        """`perm` = this epoch's shuffled GLOBAL row ids; the reference's minibatches, split by row owner."""
        if steps_per_pass or global_batch_rows is not None:
            raise ValueError("ShardedEngine.train_epoch derives the step plan itself")
        grows, per_rank = split_by_owner(perm, self.batch_size, self.bounds)
        outs = self._each("train_epoch_ragged", [(lp, lr, grows, d_steps, g_steps) for lp, lr in per_rank])
        self._dirty = True
        return outs[0]                    # global means: identical on every rank

    def train_step(self, kind, uids):
        raise L.GanmfError("ShardedEngine: single updates go through train_epoch")

    # ---- everything that needs all of both factor matrices runs on the master ------------------------------
    def scores(self, ids, transposed=False):
        self._sync_master()
        return self.master.scores(ids, transposed)

    def set_seen(self, urm_eval_csr):
        self.master.set_seen(urm_eval_csr)

    def set_score_filter(self, items_to_compute=None, mask_cold=False):
        self.master.set_score_filter(items_to_compute, mask_cold)

    def recommend(self, ids, cutoff, transposed=False, remove_seen=True):
        self._sync_master()
        return self.master.recommend(ids, cutoff, transposed, remove_seen)

    def set_test(self, urm_test_csr, gains):
        self.master.set_test(urm_test_csr, gains)

    def evaluate(self, ids, cutoffs, disc, ideal_cum, transposed=False, remove_seen=True):
        self._sync_master()
        return self.master.evaluate(ids, cutoffs, disc, ideal_cum, transposed, remove_seen)

    def snapshot_best(self):
        self._sync_master()
        self.master.snapshot_best()

    def restore_best(self):
        """best snapshot -> the master's parameters -> back to the ranks (training may go on from there)"""
        self._sync_master()
        self.master.restore_best()
        u = self.master.get_tensor(L.T_USER_EMB)
        self._each("set_tensor", [(L.T_USER_EMB, u[lo:hi]) for lo, hi in self.bounds])
        tid, tids = 0, [L.T_ITEM_EMB]
        while True:
            try:
                self.master.shape(tid)
            except L.GanmfError:
                break
            tids.append(tid)
            tid += 1
        for t in tids:
            self._all("set_tensor", t, self.master.get_tensor(t))

    def bench_scores(self, n, transposed=False, iters=10):
        self._sync_master()
        return self.master.bench_scores(n, transposed, iters)
