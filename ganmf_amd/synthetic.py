"""Synthetic user x item matrices for the bench: binary implicit feedback in CSR,
item popularity Zipf(s) over a random item permutation, user activity log-normal clipped to
[2, N/4], every row non-empty; plus the documented Glorot-uniform parameter initialisation."""
import numpy as np
import scipy.sparse as sps


def synthetic_urm(n_users, n_items, density=0.035, zipf_s=1.0, sigma=1.0, seed=1337):
    rng = np.random.RandomState(seed)
    mean_cnt = density * n_items
    mu = np.log(mean_cnt) - 0.5 * sigma ** 2
    cnt = np.clip(np.rint(rng.lognormal(mu, sigma, n_users)), 2, max(2, n_items // 4)).astype(np.int64)
    # rescale towards the target density after clipping
    cnt = np.clip(np.rint(cnt * (mean_cnt * n_users / cnt.sum())), 2, max(2, n_items // 4)).astype(np.int64)
    w = 1.0 / np.arange(1, n_items + 1) ** zipf_s
    cdf = np.cumsum(w / w.sum())
    item_of_rank = rng.permutation(n_items)
    # inverse-CDF draws with replacement, oversampled, then de-duplicated per user
    draws = (cnt * 1.6).astype(np.int64) + 16
    owner = np.repeat(np.arange(n_users), draws)
    ranks = np.minimum(np.searchsorted(cdf, rng.rand(owner.size)), n_items - 1)
    key = owner * np.int64(n_items) + item_of_rank[ranks]
    key = np.unique(key)
    owner_u = key // n_items
    # keep at most cnt[u] entries per user (a random subset: shuffle ties by a random secondary key)
    order = np.lexsort((rng.rand(key.size), owner_u))
    key, owner_u = key[order], owner_u[order]
    start = np.searchsorted(owner_u, np.arange(n_users))
    pos_in_user = np.arange(key.size) - start[owner_u]
    keep = pos_in_user < cnt[owner_u]
    key, owner_u = key[keep], owner_u[keep]
    items = (key % n_items).astype(np.int32)
    m = sps.csr_matrix((np.ones(key.size, np.float32), (owner_u, items)), shape=(n_users, n_items))
    m.sum_duplicates()
    m.sort_indices()
    assert np.all(np.diff(m.indptr) > 0)
    return m


def glorot_params(n_users, n_items, k, e, seed=1337):
    """Glorot-uniform from RandomState(seed) in tensor order We, Wd, U, V (biases zero)."""
    rng = np.random.RandomState(seed)

    def g(a, b):
        lim = np.sqrt(6.0 / (a + b))
        return rng.uniform(-lim, lim, size=(a, b)).astype(np.float32)
    return {"We": g(n_items, e), "be": np.zeros(e, np.float32), "Wd": g(e, n_items),
            "bd": np.zeros(n_items, np.float32), "U": g(n_users, k), "V": g(n_items, k)}
