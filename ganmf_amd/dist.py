"""Row-wise (user) sharding for data-parallel training on the GPUs of one node.

Every rank owns a contiguous block of the generator's users: its CSR rows, its rows of
user_embeddings and their Adam moments (never communicated).  item_embeddings and all
discriminator tensors are replicated; their gradients are all-reduced (RCCL, inside
libganmf_hip) every step, so the replicas apply identical Adam updates.

Step agreement: every rank must issue the same collectives, so all ranks run
max_r ceil(rows_r / B) steps per pass; a rank that ran out of rows contributes zero gradients.
All losses are *global* means, so each step needs the global number of rows in that slice —
computable by every rank from the shard sizes alone (no communication).
"""
import numpy as np


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
