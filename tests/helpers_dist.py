"""A stand-in for ganmf_amd.engine.Engine with the methods ShardedEngine drives, so that the host logic of the sharded
fit() -- sharding, the owner split of the reference's minibatches, the rank processes and their pipes, the gather of
rank-owned rows into the master -- is testable on CPU.  It records what it was asked to do; it computes nothing."""
import numpy as np

T_USER_EMB, T_ITEM_EMB = 100, 101


class RecordingEngine(object):
    def __init__(self, num_users, num_items, num_factors, emb_dim, batch_size, world_size=1, rank=0, row_offset=0,
                 device=0, fail_on=None, **kw):
        self.U, self.N, self.k, self.e = num_users, num_items, num_factors, emb_dim
        self.world, self.rank, self.row_offset, self.device = world_size, rank, row_offset, device
        self.t = {T_USER_EMB: np.zeros((num_users, num_factors), np.float32),
                  T_ITEM_EMB: np.zeros((num_items, num_factors), np.float32),
                  0: np.zeros((num_items, emb_dim), np.float32), 1: np.zeros((1, emb_dim), np.float32)}
        self.best = None
        self.urm = None
        self.epochs = []
        self.comm = None
        self.fail_on = fail_on
        self.closed = False

    def close(self):
        self.closed = True

    def set_urm(self, urm):
        assert urm.shape == (self.U, self.N)
        self.urm = urm

    def shape(self, tid):
        from ganmf_amd._lib import GanmfError
        if tid not in self.t:
            raise GanmfError("unknown tensor id %d" % tid)
        return self.t[tid].shape

    def set_tensor(self, tid, arr, slot=0):
        self.t[tid] = np.array(arr, np.float32).reshape(self.t[tid].shape)

    def get_tensor(self, tid, slot=0):
        return self.t[tid].copy()

    def comm_unique_id(self):
        return b"u" * 128

    def comm_init(self, uid):
        self.comm = ("rccl", bytes(uid))

    def comm_init_local(self, group):
        self.comm = ("local", group)

    def train_epoch_ragged(self, perm, local_rows, global_rows, d_steps=1, g_steps=1):
        if self.fail_on == "train":
            raise RuntimeError("synthetic failure on rank %d" % self.rank)
        perm = np.asarray(perm)
        assert perm.size == int(np.sum(local_rows)) and len(local_rows) == len(global_rows)
        assert perm.size == 0 or (perm.min() >= 0 and perm.max() < self.U)
        self.epochs.append((perm.copy(), np.array(local_rows), np.array(global_rows)))
        # "training": every row that was visited gets +1 (+ its global id / 1000 on the first visit), V counts the steps
        self.t[T_USER_EMB][perm] += 1.0
        self.t[T_ITEM_EMB] += len(global_rows)
        n = len(global_rows)
        return (np.arange(d_steps * n, dtype=np.float32), np.arange(g_steps * n, dtype=np.float32) + 0.5)

    def scores(self, ids, transposed=False):
        return self.t[T_USER_EMB][np.asarray(ids)] @ self.t[T_ITEM_EMB].T

    def set_seen(self, urm):
        self.seen = urm.shape

    def set_score_filter(self, items_to_compute=None, mask_cold=False):
        self.score_filter = (None if items_to_compute is None else list(items_to_compute), bool(mask_cold))

    def snapshot_best(self):
        self.best = {k: v.copy() for k, v in self.t.items()}

    def restore_best(self):
        self.t = {k: v.copy() for k, v in self.best.items()}

    def visited(self):
        return [e[0] for e in self.epochs]


def recording_factory(**kw):
    return RecordingEngine(**kw)


def failing_rank1_factory(**kw):
    return RecordingEngine(fail_on="train" if kw.get("rank") == 1 and kw.get("world_size", 1) > 1 else None, **kw)
