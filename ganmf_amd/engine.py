"""Thin object wrapper of one libganmf_hip handle (one GPU).  No arithmetic happens here."""
import ctypes as C

import numpy as np

from . import _lib as L


def _f32p(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _i32p(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


class Engine:
    def __init__(self, num_users, num_items, num_factors, emb_dim, batch_size, d_lr=1e-4, g_lr=1e-4, d_reg=0.0,
                 g_reg=0.0, m=1.0, recon_coefficient=1e-2, model=L.MODEL_GANMF, d_layers=1, d_act="linear",
                 device=0, world_size=1, rank=0, row_offset=0, mfma=None):
        """mfma: None/"auto" (fp32-accurate, kernel chosen per GEMM), "f32" (plain fp32 MFMA everywhere) or "bf16"
        (operands rounded to bf16, fp32 accumulate and fp32 master weights/Adam: the mixed-precision variant)."""
        self.lib = L.load_library()
        if self.lib.ganmf_device_count() < 1:
            raise L.GanmfError("no HIP device visible: libganmf_hip has no CPU fallback")
        cfg = L.Cfg(abi_version=L.ABI_VERSION, model=model, num_users=num_users, num_items=num_items,
                    num_factors=num_factors, emb_dim=emb_dim, d_layers=d_layers, d_act=L.ACT[d_act],
                    batch_size=batch_size, d_lr=d_lr, g_lr=g_lr, d_reg=d_reg, g_reg=g_reg, m=m,
                    recon_coefficient=recon_coefficient, device=device, world_size=world_size, rank=rank,
                    row_offset=row_offset, flags=L.MFMA_FLAGS[mfma])
        self.cfg = cfg
        self.h = C.c_void_p()
        L.check(self.lib.ganmf_create(C.byref(cfg), C.byref(self.h)), "ganmf_create")
        self.num_users, self.num_items = num_users, num_items
        self.batch_size = min(batch_size, num_users)

    def close(self):
        if getattr(self, "h", None) is not None and self.h.value:
            self.lib.ganmf_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- data -------------------------------------------------------------------------------
    def set_urm(self, urm_csr):
        urm = urm_csr.tocsr()
        urm.sum_duplicates()
        urm.sort_indices()
        indptr = np.ascontiguousarray(urm.indptr, dtype=np.int64)
        indices = np.ascontiguousarray(urm.indices, dtype=np.int32)
        data = np.ascontiguousarray(urm.data, dtype=np.float32)
        L.check(self.lib.ganmf_set_urm_csr(self.h, indptr.ctypes.data_as(C.POINTER(C.c_int64)), _i32p(indices),
                                           _f32p(data), urm.shape[0], urm.shape[1]), "ganmf_set_urm_csr")

    def shape(self, tid):
        r, c = C.c_int64(), C.c_int64()
        L.check(self.lib.ganmf_tensor_shape(self.h, tid, C.byref(r), C.byref(c)), "ganmf_tensor_shape")
        return r.value, c.value

    def set_tensor(self, tid, arr, slot=L.SLOT_PARAM):
        a = np.ascontiguousarray(arr, dtype=np.float32)
        L.check(self.lib.ganmf_set_tensor(self.h, tid, slot, _f32p(a), a.size), "ganmf_set_tensor")

    def get_tensor(self, tid, slot=L.SLOT_PARAM):
        r, c = self.shape(tid)
        out = np.empty((r, c), dtype=np.float32)
        L.check(self.lib.ganmf_get_tensor(self.h, tid, slot, _f32p(out), out.size), "ganmf_get_tensor")
        return out

    def adam_powers(self):
        out = np.empty(4, dtype=np.float32)
        L.check(self.lib.ganmf_get_adam_powers(self.h, _f32p(out)), "ganmf_get_adam_powers")
        return out

    def set_adam_powers(self, p):
        a = np.ascontiguousarray(p, dtype=np.float32)
        L.check(self.lib.ganmf_set_adam_powers(self.h, _f32p(a)), "ganmf_set_adam_powers")

    # -- training ---------------------------------------------------------------------------
    def train_epoch(self, perm, d_steps=1, g_steps=1, steps_per_pass=0, global_batch_rows=None):
        perm = np.ascontiguousarray(perm, dtype=np.int32)
        n = perm.size
        per_pass = max(-(-n // self.batch_size), steps_per_pass)
        dl = np.zeros(max(d_steps * per_pass, 1), dtype=np.float32)
        gl = np.zeros(max(g_steps * per_pass, 1), dtype=np.float32)
        gb = None
        if global_batch_rows is not None:
            gb = np.ascontiguousarray(global_batch_rows, dtype=np.int32)
            assert gb.size == per_pass
        L.check(self.lib.ganmf_train_epoch(self.h, _i32p(perm), n, d_steps, g_steps, steps_per_pass,
                                           _i32p(gb) if gb is not None else None, _f32p(dl), _f32p(gl)),
                "ganmf_train_epoch")
        return dl[:d_steps * per_pass], gl[:g_steps * per_pass]

    def train_epoch_ragged(self, perm, local_batch_rows, global_batch_rows, d_steps=1, g_steps=1):
        """Slices of given sizes (ganmf_train_epoch_ragged): slice i = the next local_batch_rows[i] rows of `perm`, part of a
        global minibatch of global_batch_rows[i] rows (row-sharded fit(), ganmf_amd/dist.py)."""
        perm = np.ascontiguousarray(perm, dtype=np.int32)
        lb = np.ascontiguousarray(local_batch_rows, dtype=np.int32)
        gb = np.ascontiguousarray(global_batch_rows, dtype=np.int32)
        assert lb.size == gb.size and int(lb.sum()) == perm.size
        per_pass = lb.size
        dl = np.zeros(max(d_steps * per_pass, 1), dtype=np.float32)
        gl = np.zeros(max(g_steps * per_pass, 1), dtype=np.float32)
        L.check(self.lib.ganmf_train_epoch_ragged(self.h, _i32p(perm), perm.size, d_steps, g_steps, per_pass, _i32p(gb),
                                                  _i32p(lb), _f32p(dl), _f32p(gl)), "ganmf_train_epoch_ragged")
        return dl[:d_steps * per_pass], gl[:g_steps * per_pass]

    def train_step(self, kind, uids):
        u = np.ascontiguousarray(uids, dtype=np.int32)
        loss = C.c_float()
        L.check(self.lib.ganmf_train_step(self.h, kind, _i32p(u), u.size, C.byref(loss)), "ganmf_train_step")
        return np.float32(loss.value)

    def scores(self, ids, transposed=False):
        ids = np.ascontiguousarray(ids, dtype=np.int32).ravel()
        width = self.num_users if transposed else self.num_items
        out = np.empty((ids.size, width), dtype=np.float32)
        if ids.size:
            L.check(self.lib.ganmf_scores(self.h, _i32p(ids), ids.size, int(transposed), _f32p(out)), "ganmf_scores")
        return out

    def set_seen(self, urm_eval_csr):
        """URM_train in evaluation orientation (rows = users the evaluator asks about)."""
        urm = urm_eval_csr.tocsr()
        urm.sort_indices()
        indptr = np.ascontiguousarray(urm.indptr, dtype=np.int64)
        indices = np.ascontiguousarray(urm.indices, dtype=np.int32)
        L.check(self.lib.ganmf_set_seen_csr(self.h, indptr.ctypes.data_as(C.POINTER(C.c_int64)), _i32p(indices),
                                            urm.shape[0], urm.shape[1]), "ganmf_set_seen_csr")

    def set_score_filter(self, items_to_compute=None, mask_cold=False):
        """MF contract (BaseMatrixFactorizationRecommender.py:113-119,128-143) for every later scores / recommend / evaluate
        call: only `items_to_compute` keep their scores (the others -inf); rows that are empty in the set_seen() matrix score
        -inf everywhere."""
        if items_to_compute is None or len(items_to_compute) == 0:
            L.check(self.lib.ganmf_set_score_filter(self.h, None, 0, int(bool(mask_cold))), "ganmf_set_score_filter")
            return
        items = np.ascontiguousarray(np.asarray(items_to_compute).reshape(-1), dtype=np.int32)
        L.check(self.lib.ganmf_set_score_filter(self.h, _i32p(items), len(items), int(bool(mask_cold))), "ganmf_set_score_filter")

    def recommend(self, ids, cutoff, transposed=False, remove_seen=True):
        """device top-k: returns (items [n, cutoff] int32 with -1 padding, scores [n, cutoff])"""
        ids = np.ascontiguousarray(ids, dtype=np.int32).ravel()
        items = np.empty((ids.size, cutoff), dtype=np.int32)
        vals = np.empty((ids.size, cutoff), dtype=np.float32)
        if ids.size:
            L.check(self.lib.ganmf_recommend(self.h, _i32p(ids), ids.size, int(transposed), int(cutoff), int(remove_seen),
                                             _i32p(items), _f32p(vals)), "ganmf_recommend")
        return items, vals

    def set_test(self, urm_test_csr, gains):
        """URM_test in evaluation orientation, column indices sorted inside each row; `gains` = 2^rating - 1 per stored
        entry (float64, same order as urm_test_csr.data)."""
        urm = urm_test_csr.tocsr()
        assert urm.has_sorted_indices
        indptr = np.ascontiguousarray(urm.indptr, dtype=np.int64)
        indices = np.ascontiguousarray(urm.indices, dtype=np.int32)
        g = np.ascontiguousarray(gains, dtype=np.float64)
        assert g.size == indices.size
        dp = C.POINTER(C.c_double)
        L.check(self.lib.ganmf_set_test_csr(self.h, indptr.ctypes.data_as(C.POINTER(C.c_int64)), _i32p(indices),
                                            g.ctypes.data_as(dp), urm.shape[0], urm.shape[1]), "ganmf_set_test_csr")

    def evaluate(self, ids, cutoffs, disc, ideal_cum, transposed=False, remove_seen=True):
        """Sums over the users `ids` of the nine ranking metrics (L.EVAL_METRICS) per cut-off, formed on the device from
        the device's own top-k lists: returns a [len(cutoffs), 9] float64 array."""
        ids = np.ascontiguousarray(ids, dtype=np.int32).ravel()
        cut = np.ascontiguousarray(cutoffs, dtype=np.int32).ravel()
        K = int(cut.max())
        disc = np.ascontiguousarray(disc, dtype=np.float64).ravel()
        ideal = np.ascontiguousarray(ideal_cum, dtype=np.float64)
        assert disc.size >= K and ideal.shape == (ids.size, K)
        out = np.zeros((cut.size, len(L.EVAL_METRICS)), dtype=np.float64)
        dp = C.POINTER(C.c_double)
        if ids.size:
            L.check(self.lib.ganmf_evaluate(self.h, _i32p(ids), ids.size, int(transposed), int(remove_seen), _i32p(cut), cut.size,
                                            disc.ctypes.data_as(dp), ideal.ctypes.data_as(dp), out.ctypes.data_as(dp)),
                    "ganmf_evaluate")
        return out

    def snapshot_best(self):
        L.check(self.lib.ganmf_snapshot_best(self.h), "ganmf_snapshot_best")

    def restore_best(self):
        L.check(self.lib.ganmf_restore_best(self.h), "ganmf_restore_best")

    # -- measurement ------------------------------------------------------------------------
    def timer_start(self):
        """hipEvent on the library's stream in front of whatever is enqueued next (bench.py's timed region)."""
        L.check(self.lib.ganmf_stream_timer(self.h, 0, None), "ganmf_stream_timer")

    def timer_stop(self):
        """milliseconds the stream spent since timer_start (waits for the stream)."""
        ms = C.c_double(0.0)
        L.check(self.lib.ganmf_stream_timer(self.h, 1, C.byref(ms)), "ganmf_stream_timer")
        return float(ms.value)

    def profile(self, on):
        L.check(self.lib.ganmf_profile_enable(self.h, int(on)), "ganmf_profile_enable")

    def profile_read(self):
        buf = (L.ProfEntry * L.PROF_MAX)()
        n = C.c_int32()
        L.check(self.lib.ganmf_profile_read(self.h, buf, L.PROF_MAX, C.byref(n)), "ganmf_profile_read")
        return [dict(name=buf[i].name.decode(), launches=buf[i].launches, ms=buf[i].ms, flops=buf[i].flops,
                     bytes=buf[i].bytes) for i in range(n.value)]

    def bench_scores(self, n, transposed=False, iters=10):
        ms = C.c_float()
        L.check(self.lib.ganmf_bench_scores(self.h, n, int(transposed), iters, C.byref(ms)), "ganmf_bench_scores")
        return ms.value

    def comm_info(self):
        """(ranks in the communicator as RCCL reports them, this handle's rank); (0, -1) without a communicator"""
        w, r = C.c_int32(0), C.c_int32(-1)
        L.check(self.lib.ganmf_comm_info(self.h, C.byref(w), C.byref(r)), "ganmf_comm_info")
        return int(w.value), int(r.value)

    def comm_abort(self):
        """end this engine's communicator from ANOTHER thread than the one inside a training call (ganmf_comm_abort): the
        call in flight returns with an error instead of waiting for a peer that failed; close() is what is left to do"""
        if self.h:
            L.check(self.lib.ganmf_comm_abort(self.h), "ganmf_comm_abort")

    def comm_init_local(self, group_id):
        """join the in-process loopback communicator `group_id` (all world_size engines of this process must)"""
        L.check(self.lib.ganmf_comm_init_local(self.h, int(group_id)), "ganmf_comm_init_local")

    def comm_unique_id(self):
        """128 bytes rank 0 hands to every rank's comm_init (ganmf_comm_unique_id)"""
        return comm_unique_id()

    def comm_init(self, id_bytes):
        arr = (C.c_uint8 * 128).from_buffer_copy(bytes(id_bytes))
        L.check(self.lib.ganmf_comm_init(self.h, arr), "ganmf_comm_init")


def comm_unique_id():
    lib = L.load_library()
    arr = (C.c_uint8 * 128)()
    L.check(lib.ganmf_comm_unique_id(arr), "ganmf_comm_unique_id")
    return bytes(arr)


def gemm_f32(A, B, a_kmajor=False, b_kmajor=False, tile=0, nsplit=0, iters=1, device=0):
    """C = op(A).op(B) through the stand-alone entry (tests / bench)."""
    lib = L.load_library()
    A = np.ascontiguousarray(A, dtype=np.float32)
    B = np.ascontiguousarray(B, dtype=np.float32)
    K, M = (A.shape if a_kmajor else A.shape[::-1])
    Kb, N = (B.shape if b_kmajor else B.shape[::-1])
    assert K == Kb, (A.shape, B.shape)
    out = np.empty((M, N), dtype=np.float32)
    ms = C.c_float()
    L.check(lib.ganmf_gemm_f32(device, _f32p(A), _f32p(B), _f32p(out), M, N, K, int(a_kmajor), int(b_kmajor),
                               tile, nsplit, iters, C.byref(ms)), "ganmf_gemm_f32")
    return out, ms.value
