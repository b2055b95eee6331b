"""CPU oracle for the GANMF / DisGANMF training hot path.  TEST INFRASTRUCTURE ONLY.

This file is a numpy restatement of the arithmetic that the reference executes through
TensorFlow 1.12 in ``GANRec/GANMF.py`` and ``GANRec/DisGANMF.py``.  It is the checker the
HIP path is compared against.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it; the product (``ganmf_amd``) never does.

PARITY STATUS
    * scoring / recommend / evaluation: PINNED by the reference's surviving checkpoint
      (KAT-1, ``tests/golden/kat1_*``; see ``oracle/make_golden.py``).
    * training arithmetic: the reference has no test of ``GANRec/*`` and TensorFlow 1.12 is
      not installable here, so the per-step arithmetic is **parity unpinned** by reference
      tests.  It is anchored instead on (a) gradient checks of every hand-written backward
      formula against ``torch.autograd`` on the literal loss expressions of
      ``GANMF.py:131-135`` / ``DisGANMF.py:114-136`` (``tests/test_oracle_grads.py``), and
      (b) end-to-end statistical known answers: training with the reference's tuned
      hyper-parameters reproduces the published MAP@5 (``tests/golden/statistical_kat.json``).

Reference sites followed (all paths relative to /root/reference):
    graph                GANRec/GANMF.py:62-84      autoencoder() / generator()
    losses               GANRec/GANMF.py:131-135
    update ops           GANRec/GANMF.py:104-105,138-139   two tf.train.AdamOptimizer
    schedule             GANRec/GANMF.py:156-203    shuffle once, d_steps passes, g_steps passes
    scoring              GANRec/GANMF.py:285-292
    DisGANMF graph       GANRec/DisGANMF.py:57-79
    DisGANMF losses      GANRec/DisGANMF.py:110-140

Third-party arithmetic (tensorflow==1.12.0, pip_requirements.txt:12, source absent) restated
from its published kernels:
    tf.layers.dense                y = x @ kernel + bias                (kernel [in, units])
    tf.losses.mean_squared_error   sum((pred - labels)^2) / numel, gradient flows to BOTH args
    tf.nn.l2_loss                  sum(v^2) / 2
    tf.maximum(0, h)               gradient to h only where h > 0
    sigmoid_cross_entropy_with_logits(z, x) = max(x,0) - x*z + log1p(exp(-|x|))
    ApplyAdam (dense)              alpha = lr*sqrt(1-b2p)/(1-b1p); m += (g-m)*(1-b1);
                                   v += (g*g-v)*(1-b2); var -= (m*alpha)/(sqrt(v)+eps)
    AdamOptimizer._apply_sparse_shared (user_embeddings: IndexedSlices gradient aggregated
                                   with the dense g_reg*l2 gradient -> every row is an index)
                                   m = m*b1 + g*(1-b1); v = v*b2 + g*g*(1-b2);
                                   var -= alpha*m/(sqrt(v)+eps)
    beta powers                    fp32 variables starting at b1, b2; multiplied by b1, b2
                                   after every minimize() run (AdamOptimizer._finish)
"""
from __future__ import annotations

import numpy as np

BETA1 = 0.9
BETA2 = 0.999
EPSILON = 1e-8

ACTIVATIONS = ("linear", "tanh", "relu", "sigmoid")


def glorot_uniform(rng: np.random.RandomState, shape, dtype=np.float32):
    """tf.glorot_uniform_initializer for a 2-D variable [a, b]: U(-L, L), L = sqrt(6/(a+b)).
    (GANMF.py:57).  TF's Philox stream is not reproducible without TF, so parity tests always
    inject explicit weights; this is the build's own documented default."""
    fan_in, fan_out = shape
    limit = np.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-limit, limit, size=shape).astype(dtype)


class _Adam:
    """One tf.train.AdamOptimizer instance: a shared (beta1_power, beta2_power) pair and
    per-variable (m, v) slots."""

    def __init__(self, lr, dtype):
        self.dt = np.dtype(dtype).type
        self.lr = self.dt(lr)
        self.b1 = self.dt(BETA1)
        self.b2 = self.dt(BETA2)
        self.eps = self.dt(EPSILON)
        self.b1p = self.dt(BETA1)
        self.b2p = self.dt(BETA2)
        self.slots = {}

    def _slot(self, name, var):
        if name not in self.slots:
            self.slots[name] = (np.zeros_like(var), np.zeros_like(var))
        return self.slots[name]

    def alpha(self):
        one = self.dt(1)
        return self.dt(self.lr * np.sqrt(one - self.b2p) / (one - self.b1p))

    def apply_dense(self, name, var, grad):
        m, v = self._slot(name, var)
        one = self.dt(1)
        a = self.alpha()
        m += (grad - m) * (one - self.b1)
        v += (grad * grad - v) * (one - self.b2)
        var -= (m * a) / (np.sqrt(v) + self.eps)

    def apply_sparse_all_rows(self, name, var, grad):
        """_apply_sparse_shared with every row present as an index (see module docstring)."""
        m, v = self._slot(name, var)
        one = self.dt(1)
        a = self.alpha()
        m *= self.b1
        m += grad * (one - self.b1)
        v *= self.b2
        v += (grad * grad) * (one - self.b2)
        var -= a * m / (np.sqrt(v) + self.eps)

    def finish(self):
        self.b1p = self.dt(self.b1p * self.b1)
        self.b2p = self.dt(self.b2p * self.b2)


def _densify(urm_csr, uids, dtype):
    """real_histories = URM_train[uids].toarray()   (GANMF.py:183-184)"""
    return np.asarray(urm_csr[uids].toarray(), dtype=dtype)


def batch_slices(n_users, batch_size):
    """The reference's ragged slicing loop (GANMF.py:177-189)."""
    out = []
    start = 0
    while start < n_users:
        end = min(start + batch_size, n_users)
        out.append((start, end))
        start = end
    return out


class GANMFOracle:
    """MF generator + linear auto-encoder discriminator (GANMF.py:53-139)."""

    D_NAMES = ("We", "be", "Wd", "bd")   # autoencoder/encoding/{kernel,bias}, decoding/{kernel,bias}
    G_NAMES = ("U", "V")                 # generator/{user,item}_embeddings

    def __init__(self, num_users, num_items, num_factors=10, emb_dim=32, d_lr=1e-4, g_lr=1e-4,
                 d_reg=0.0, g_reg=0.0, m=1.0, recon_coefficient=1e-2, dtype=np.float32, seed=1337):
        self.nu, self.ni, self.k, self.e = num_users, num_items, num_factors, emb_dim
        self.dtype = np.dtype(dtype)
        self.dt = self.dtype.type
        self.d_reg, self.g_reg = self.dt(d_reg), self.dt(g_reg)
        self.m, self.alpha = self.dt(m), self.dt(recon_coefficient)
        self.opt_d = _Adam(d_lr, dtype)
        self.opt_g = _Adam(g_lr, dtype)
        rng = np.random.RandomState(seed)
        # tensor order We, be, Wd, bd, U, V ; always drawn in float32 then widened
        We = glorot_uniform(rng, (num_items, emb_dim))
        Wd = glorot_uniform(rng, (emb_dim, num_items))
        U = glorot_uniform(rng, (num_users, num_factors))
        V = glorot_uniform(rng, (num_items, num_factors))
        self.p = {
            "We": We.astype(dtype), "be": np.zeros(emb_dim, dtype),
            "Wd": Wd.astype(dtype), "bd": np.zeros(num_items, dtype),
            "U": U.astype(dtype), "V": V.astype(dtype),
        }
        self.hinge_active_last = None

    # -- parameter access -------------------------------------------------------------
    def set_params(self, **kw):
        for name, val in kw.items():
            assert self.p[name].shape == np.shape(val), (name, self.p[name].shape, np.shape(val))
            self.p[name] = np.array(val, dtype=self.dtype)

    def get_params(self):
        return {k: v.copy() for k, v in self.p.items()}

    # -- forward pieces ------------------------------------------------------------------
    def generator(self, uids):
        """fake = U[uids] @ V.T  (GANMF.py:82-83)"""
        return self.p["U"][uids] @ self.p["V"].T

    def autoencoder(self, inp):
        """(GANMF.py:62-70) returns encoding, delta = recon - inp, loss."""
        E = inp @ self.p["We"] + self.p["be"]
        R = E @ self.p["Wd"] + self.p["bd"]
        delta = R - inp
        loss = self.dt(np.sum(delta * delta) / self.dt(delta.size))
        return E, delta, loss

    def l2(self, names):
        return self.dt(sum(np.sum(self.p[n] * self.p[n]) for n in names) / self.dt(2))

    # -- one discriminator update (GANMF.py:131-132,138,186-187) ---------------------------
    def d_grads(self, uids, X):
        B, N = X.shape
        F = self.generator(uids)                       # constant w.r.t. theta_D
        Er, dr, Lr = self.autoencoder(X)
        Ef, df, Lf = self.autoencoder(F)
        h = self.dt(self.m * Lr - Lf)
        loss = self.dt(Lr + max(self.dt(0), h) + self.d_reg * self.l2(self.D_NAMES))
        active = bool(h > 0)
        self.hinge_active_last = active
        cr = self.dt(1) + (self.m if active else self.dt(0))
        cf = self.dt(-1) if active else self.dt(0)
        s = self.dt(2) / self.dt(B * N)
        g = {n: np.zeros_like(self.p[n]) for n in self.D_NAMES}
        for inp, E, delta, c in ((X, Er, dr, cr), (F, Ef, df, cf)):
            dR = (c * s) * delta
            g["Wd"] += E.T @ dR
            g["bd"] += dR.sum(axis=0)
            dE = dR @ self.p["Wd"].T
            g["We"] += inp.T @ dE
            g["be"] += dE.sum(axis=0)
        for n in self.D_NAMES:
            g[n] += self.d_reg * self.p[n]
        return loss, g

    def d_step(self, uids, X):
        X = np.asarray(X, dtype=self.dtype)
        loss, g = self.d_grads(uids, X)
        for n in self.D_NAMES:
            self.opt_d.apply_dense(n, self.p[n], g[n])
        self.opt_d.finish()
        return loss

    # -- one generator update (GANMF.py:133-135,139,200-201) -------------------------------
    def g_grads(self, uids, X):
        B, N = X.shape
        e = self.e
        Ub = self.p["U"][uids]
        F = Ub @ self.p["V"].T
        Er = X @ self.p["We"] + self.p["be"]
        Ef, df, Lf = self.autoencoder(F)
        dfm = Ef - Er
        fm = self.dt(np.sum(dfm * dfm) / self.dt(B * e))
        loss = self.dt((self.dt(1) - self.alpha) * Lf + self.alpha * fm + self.g_reg * self.l2(self.G_NAMES))
        dR = ((self.dt(1) - self.alpha) * self.dt(2) / self.dt(B * N)) * df
        dE = dR @ self.p["Wd"].T + (self.alpha * self.dt(2) / self.dt(B * e)) * dfm
        dF = dE @ self.p["We"].T - dR               # MSE gradient reaches F through both args
        gUb = dF @ self.p["V"]
        gV = dF.T @ Ub
        gU = np.array(self.g_reg * self.p["U"], dtype=self.dtype)   # dense l2 term, all rows
        gU[uids] += gUb                             # uids unique inside a batch
        gV = gV + self.g_reg * self.p["V"]
        return loss, {"U": gU, "V": gV}

    def g_step(self, uids, X):
        X = np.asarray(X, dtype=self.dtype)
        loss, g = self.g_grads(uids, X)
        self.opt_g.apply_sparse_all_rows("U", self.p["U"], g["U"])
        self.opt_g.apply_dense("V", self.p["V"], g["V"])
        self.opt_g.finish()
        return loss

    # -- epoch schedule (GANMF.py:172-203) ------------------------------------------------
    def train_epoch(self, urm_csr, perm, batch_size, d_steps=1, g_steps=1):
        """`perm` is the already shuffled all_users array of this epoch."""
        dl, gl = [], []
        slices = batch_slices(len(perm), batch_size)
        for _ in range(d_steps):
            for a, b in slices:
                uids = perm[a:b]
                dl.append(self.d_step(uids, _densify(urm_csr, uids, self.dtype)))
        for _ in range(g_steps):
            for a, b in slices:
                uids = perm[a:b]
                gl.append(self.g_step(uids, _densify(urm_csr, uids, self.dtype)))
        return np.array(dl, dtype=self.dtype), np.array(gl, dtype=self.dtype)

    # -- scoring (GANMF.py:285-292) -----------------------------------------------------
    def scores(self, ids, item_mode=False):
        if item_mode:
            return (self.p["U"] @ self.p["V"].T).T[ids]
        return self.p["U"][ids] @ self.p["V"].T


def _act(name, z):
    if name == "linear":
        return z
    if name == "tanh":
        return np.tanh(z)
    if name == "relu":
        return np.maximum(z, 0)
    if name == "sigmoid":
        return 1 / (1 + np.exp(-z))
    raise ValueError(name)


def _act_grad(name, z, a):
    """derivative of the activation given pre-activation z and output a"""
    if name == "linear":
        return np.ones_like(z)
    if name == "tanh":
        return 1 - a * a
    if name == "relu":
        return (z > 0).astype(z.dtype)
    if name == "sigmoid":
        return a * (1 - a)
    raise ValueError(name)


def _sce(z, x):
    """tf.nn.sigmoid_cross_entropy_with_logits(labels=z, logits=x)"""
    return np.maximum(x, 0) - x * z + np.log1p(np.exp(-np.abs(x)))


def _sigmoid(x):
    return 1 / (1 + np.exp(-x))


class DisGANMFOracle:
    """Binary-classifier discriminator variant (DisGANMF.py:57-79,110-140).

    D input is concat([float(uid), profile]) (DisGANMF.py:59,110-111) so layer_0's kernel has
    num_items+1 rows, row 0 multiplying the raw user id.  As written in the reference the
    generator *minimises* loss_fake (DisGANMF.py:135); reproduced as is."""

    def __init__(self, num_users, num_items, num_factors=10, d_layers=1, d_nodes=32,
                 d_hidden_act="linear", d_lr=1e-4, g_lr=1e-4, d_reg=0.0, g_reg=0.0,
                 recon_coefficient=1e-2, dtype=np.float32, seed=1337):
        assert d_hidden_act in ACTIVATIONS
        self.nu, self.ni, self.k = num_users, num_items, num_factors
        self.L, self.e, self.act = d_layers, d_nodes, d_hidden_act
        self.dtype = np.dtype(dtype)
        self.dt = self.dtype.type
        self.d_reg, self.g_reg, self.alpha = self.dt(d_reg), self.dt(g_reg), self.dt(recon_coefficient)
        self.opt_d = _Adam(d_lr, dtype)
        self.opt_g = _Adam(g_lr, dtype)
        rng = np.random.RandomState(seed)
        self.p = {}
        fan_in = num_items + 1
        for l in range(d_layers):
            self.p["W%d" % l] = glorot_uniform(rng, (fan_in, d_nodes)).astype(dtype)
            self.p["b%d" % l] = np.zeros(d_nodes, dtype)
            fan_in = d_nodes
        self.p["Wo"] = glorot_uniform(rng, (fan_in, 1)).astype(dtype)
        self.p["bo"] = np.zeros(1, dtype)
        self.p["U"] = glorot_uniform(rng, (num_users, num_factors)).astype(dtype)
        self.p["V"] = glorot_uniform(rng, (num_items, num_factors)).astype(dtype)
        self.D_NAMES = tuple(n for l in range(d_layers) for n in ("W%d" % l, "b%d" % l)) + ("Wo", "bo")
        self.G_NAMES = ("U", "V")

    set_params = GANMFOracle.set_params
    get_params = GANMFOracle.get_params
    l2 = GANMFOracle.l2

    def discriminator(self, uids, inp):
        """returns list of (h_in, z, a) per hidden layer, features, logits [B]"""
        h = np.concatenate([np.asarray(uids, dtype=self.dtype).reshape(-1, 1), inp], axis=1)
        cache = []
        for l in range(self.L):
            z = h @ self.p["W%d" % l] + self.p["b%d" % l]
            a = _act(self.act, z)
            cache.append((h, z, a))
            h = a
        logit = (h @ self.p["Wo"] + self.p["bo"])[:, 0]
        return cache, h, logit

    def _backprop_hidden(self, cache, dh, g=None):
        """chain dh (gradient w.r.t. features) back to the layer-0 input; optionally
        accumulate parameter gradients into g."""
        for l in reversed(range(self.L)):
            h_in, z, a = cache[l]
            dz = dh * _act_grad(self.act, z, a)
            if g is not None:
                g["W%d" % l] += h_in.T @ dz
                g["b%d" % l] += dz.sum(axis=0)
            dh = dz @ self.p["W%d" % l].T
        return dh

    def d_grads(self, uids, X):
        B = X.shape[0]
        F = self.p["U"][uids] @ self.p["V"].T
        g = {n: np.zeros_like(self.p[n]) for n in self.D_NAMES}
        total = self.dt(0)
        for inp, lab in ((X, self.dt(1)), (F, self.dt(0))):
            cache, feat, logit = self.discriminator(uids, inp)
            total = total + self.dt(np.mean(_sce(lab, logit)))
            dlogit = ((_sigmoid(logit) - lab) / self.dt(B)).astype(self.dtype)
            g["Wo"] += feat.T @ dlogit[:, None]
            g["bo"] += dlogit.sum(keepdims=True)
            dh = dlogit[:, None] @ self.p["Wo"].T
            self._backprop_hidden(cache, dh, g)
        loss = self.dt(total + self.d_reg * self.l2(self.D_NAMES))
        for n in self.D_NAMES:
            g[n] += self.d_reg * self.p[n]
        return loss, g

    def d_step(self, uids, X):
        X = np.asarray(X, dtype=self.dtype)
        loss, g = self.d_grads(uids, X)
        for n in self.D_NAMES:
            self.opt_d.apply_dense(n, self.p[n], g[n])
        self.opt_d.finish()
        return loss

    def g_grads(self, uids, X):
        B = X.shape[0]
        Ub = self.p["U"][uids]
        F = Ub @ self.p["V"].T
        _, feat_r, _ = self.discriminator(uids, X)
        cache, feat_f, logit_f = self.discriminator(uids, F)
        loss_fake = self.dt(np.mean(_sce(self.dt(0), logit_f)))
        dfm = feat_f - feat_r
        fm = self.dt(np.sum(dfm * dfm) / self.dt(dfm.size))
        loss = self.dt(loss_fake + self.alpha * fm + self.g_reg * self.l2(self.G_NAMES))
        dlogit = (_sigmoid(logit_f) / self.dt(B)).astype(self.dtype)
        dh = dlogit[:, None] @ self.p["Wo"].T + (self.alpha * self.dt(2) / self.dt(dfm.size)) * dfm
        dh0 = self._backprop_hidden(cache, dh)
        dF = dh0[:, 1:]                              # drop the uid column
        gUb = dF @ self.p["V"]
        gV = dF.T @ Ub + self.g_reg * self.p["V"]
        gU = np.array(self.g_reg * self.p["U"], dtype=self.dtype)
        gU[uids] += gUb
        return loss, {"U": gU, "V": gV}

    def g_step(self, uids, X):
        X = np.asarray(X, dtype=self.dtype)
        loss, g = self.g_grads(uids, X)
        self.opt_g.apply_sparse_all_rows("U", self.p["U"], g["U"])
        self.opt_g.apply_dense("V", self.p["V"], g["V"])
        self.opt_g.finish()
        return loss

    train_epoch = GANMFOracle.train_epoch
    scores = GANMFOracle.scores


def reference_epoch_permutations(num_users, epochs, seed):
    """The reference's schedule stream: RecSysExp.set_seed does np.random.seed(seed)
    (RecSysExp.py:104-108); fit() then shuffles the SAME all_users array in place once per
    epoch (GANMF.py:156,175), so permutations compose across epochs."""
    st = np.random.RandomState(seed)   # legacy MT19937, identical stream to np.random.seed
    all_users = np.arange(num_users)
    out = []
    for _ in range(epochs):
        st.shuffle(all_users)
        out.append(all_users.copy())
    return out
