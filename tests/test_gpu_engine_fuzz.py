"""Randomised engine configurations (shapes, batch sizes larger / smaller than the matrix, hyper-parameters with the
hinge on either side, d_steps / g_steps, activations and depths for DisGANMF) — two epochs through the C ABI against
the fp64 oracle.  Complements the fixed cases of test_gpu_parity.py / test_gpu_configs.py."""
import numpy as np
import pytest
import scipy.sparse as sps

from oracle.ganmf_oracle import DisGANMFOracle, GANMFOracle

pytestmark = pytest.mark.gpu


def _err(got, ref):
    return np.max(np.abs(np.asarray(got, np.float64).reshape(np.shape(ref)) - ref)) / (np.max(np.abs(ref)) + 1e-30)


def _urm(rng, U, N):
    dens = rng.choice([0.02, 0.1, 0.4])
    m = (rng.rand(U, N) < dens).astype(np.float32)
    m[rng.rand(U) < 0.15] = 0.0                       # cold rows
    return sps.csr_matrix(m)


@pytest.mark.parametrize("seed", range(16))
def test_ganmf_random_config(seed):
    from ganmf_amd.engine import Engine
    rng = np.random.RandomState(1000 + seed)
    U, N = int(rng.randint(3, 400)), int(rng.randint(2, 700))
    k, e, B = int(rng.randint(1, 70)), int(rng.randint(1, 140)), int(rng.choice([1, 2, 7, 32, 64, 129, 500]))
    hp = dict(d_lr=float(10 ** rng.uniform(-4, -2.5)), g_lr=float(10 ** rng.uniform(-4, -2.5)),
              d_reg=float(rng.choice([0.0, 1e-4, 1e-2])), g_reg=float(rng.choice([0.0, 1e-3])),
              m=float(rng.choice([0.001, 1.0, 10.0])), recon_coefficient=float(rng.uniform(0, 1)))
    d_steps, g_steps = int(rng.randint(1, 3)), int(rng.randint(1, 3))
    urm = _urm(rng, U, N)
    o = GANMFOracle(U, N, k, e, dtype=np.float64, seed=seed, **hp)
    o.set_params(be=rng.randn(e) * 0.01, bd=rng.randn(N) * 0.01)
    eng = Engine(U, N, k, e, B, **hp)
    eng.set_urm(urm)
    ids = {"We": 0, "be": 1, "Wd": 2, "bd": 3, "U": 100, "V": 101}
    for n, tid in ids.items():
        eng.set_tensor(tid, o.p[n])
    for _ in range(2):
        perm = rng.permutation(U)
        dl_ref, gl_ref = o.train_epoch(urm, perm, min(B, U), d_steps, g_steps)
        dl, gl = eng.train_epoch(perm, d_steps, g_steps)
        np.testing.assert_allclose(dl, dl_ref, rtol=2e-4, atol=1e-7, err_msg=str((U, N, k, e, B, hp)))
        np.testing.assert_allclose(gl, gl_ref, rtol=2e-4, atol=1e-7, err_msg=str((U, N, k, e, B, hp)))
    for n, tid in ids.items():
        assert _err(eng.get_tensor(tid), o.p[n]) <= 2e-4, (n, U, N, k, e, B, hp)
    users = rng.permutation(U)[:min(U, 50)]
    assert _err(eng.scores(users), o.scores(users)) <= 2e-4
    assert _err(eng.scores(np.arange(min(N, 40)), transposed=True), o.scores(np.arange(U))[:, :min(N, 40)].T) <= 2e-4
    eng.close()


@pytest.mark.parametrize("seed", range(10))
def test_disganmf_random_config(seed):
    from ganmf_amd import _lib as L
    from ganmf_amd.engine import Engine
    rng = np.random.RandomState(2000 + seed)
    U, N = int(rng.randint(3, 90)), int(rng.randint(2, 300))      # float(uid) feeds the net: keep logits moderate
    k, e, B = int(rng.randint(1, 40)), int(rng.randint(1, 70)), int(rng.choice([1, 5, 32, 64, 200]))
    layers, act = int(rng.randint(1, 4)), str(rng.choice(["linear", "tanh", "relu", "sigmoid"]))
    hp = dict(d_lr=float(10 ** rng.uniform(-4, -3)), g_lr=float(10 ** rng.uniform(-4, -3)),
              d_reg=float(rng.choice([0.0, 1e-4])), g_reg=0.0, recon_coefficient=float(rng.uniform(0, 1)))
    urm = _urm(rng, U, N)
    o = DisGANMFOracle(U, N, k, d_layers=layers, d_nodes=e, d_hidden_act=act, dtype=np.float64, seed=seed, **hp)
    eng = Engine(U, N, k, e, B, model=L.MODEL_DISGANMF, d_layers=layers, d_act=act, m=0.0, **hp)
    eng.set_urm(urm)
    ids = {}
    for l in range(layers):
        ids["W%d" % l], ids["b%d" % l] = 2 * l, 2 * l + 1
    ids.update({"Wo": 2 * layers, "bo": 2 * layers + 1, "U": 100, "V": 101})
    for n, tid in ids.items():
        eng.set_tensor(tid, o.p[n])
    perm = rng.permutation(U)
    dl_ref, gl_ref = o.train_epoch(urm, perm, min(B, U), 1, 1)
    dl, gl = eng.train_epoch(perm, 1, 1)
    np.testing.assert_allclose(dl, dl_ref, rtol=5e-4, atol=1e-6, err_msg=str((U, N, k, e, B, layers, act)))
    np.testing.assert_allclose(gl, gl_ref, rtol=5e-4, atol=1e-6, err_msg=str((U, N, k, e, B, layers, act)))
    for n, tid in ids.items():
        assert _err(eng.get_tensor(tid), o.p[n]) <= 5e-4, (n, U, N, k, e, B, layers, act)
    eng.close()


def test_engine_create_destroy_does_not_leak():
    """The tuner creates and destroys an engine per trial: device memory must return to the pool.
    (hipMemGetInfo through the HIP runtime the library already loaded -- no second framework's device initialisation in
    the middle of a test process.)"""
    import ctypes
    from ganmf_amd.engine import Engine
    try:
        hip = ctypes.CDLL("libamdhip64.so")
    except OSError:
        hip = ctypes.CDLL("/opt/rocm/lib/libamdhip64.so")

    def free_bytes():
        assert hip.hipDeviceSynchronize() == 0
        free, total = ctypes.c_size_t(), ctypes.c_size_t()
        assert hip.hipMemGetInfo(ctypes.byref(free), ctypes.byref(total)) == 0
        return free.value
    rng = np.random.RandomState(0)
    urm = sps.csr_matrix((rng.rand(300, 500) < 0.05).astype(np.float32))

    def cycle(n):
        for _ in range(n):
            eng = Engine(300, 500, 16, 24, 64)
            eng.set_urm(urm)
            eng.set_seen(urm)
            eng.train_epoch(rng.permutation(300), 1, 1)
            eng.scores(np.arange(300))
            eng.recommend(np.arange(300), 5)
            eng.snapshot_best()
            eng.close()
    cycle(40)                                  # warm allocator / RCCL / HIP module state: what earlier tests of the same
    free0 = free_bytes()                       # process left in the runtime's pools settles here, a leak keeps growing
    cycle(40)
    free1 = free_bytes()
    assert free0 - free1 < 8 << 20, "device memory shrank by %.1f MiB over 40 engine lifetimes" % ((free0 - free1) / 2 ** 20)


@pytest.mark.parametrize("seed", range(8))
def test_low_precision_launch_forms_random_config(seed, monkeypatch):
    """Handles created with mfma = "f16" | "bf16" over random shapes wide enough for the combined launches (round 6: front_lp_kernel, pair_lp_kernel, the
    one-piece 64 x 32 kernel, DisGANMF's dz_0 rows + column sums inside the gradient GEMM's launch):
    * GANMF, two epochs: gUb + gV in one launch (pair_lp_kernel; the per-pass update of U with it) against the two stand-alone products -- bit for bit
      (GANMF_MULTI = 127 against 65, i.e. the generator launch combined on both sides: with that bit off the planner may split the generator product along K,
      another sum; bf16w = 0 on both sides for the same reason);
    * both models, one D and one G step from the same weights: the default plan against one kernel per piece (+ the round-5 kernels of DisGANMF's backward
      top) within the single-rounding error of the mode on every first moment, i.e. on every gradient."""
    from ganmf_amd import _lib as L
    from ganmf_amd.engine import Engine
    rng = np.random.RandomState(3000 + seed)
    mfma = "f16" if seed % 2 == 0 else "bf16"
    dis = seed % 4 >= 2
    U, N = int(rng.randint(300, 900)), int(rng.randint(600, 4000))
    k, e, B = int(rng.choice([16, 64, 100, 250])), int(rng.choice([64, 200, 512, 1024])), int(rng.choice([64, 96, 128]))
    hp = dict(d_lr=1e-4, g_lr=2e-4, d_reg=float(rng.choice([0.0, 1e-4])), g_reg=0.0, recon_coefficient=float(rng.uniform(0.05, 0.9)))
    urm = sps.csr_matrix((rng.rand(U, N) < 0.04).astype(np.float32))
    if dis:
        w = {"W0": rng.randn(N + 1, e) * 0.03, "b0": rng.randn(e) * 0.01, "Wo": rng.randn(e, 1) * 0.1, "bo": np.zeros(1), "U": rng.randn(U, k) * 0.1, "V": rng.randn(N, k) * 0.1}
        w["W0"][0, :] *= 1.0 / U      # the float(uid) row
        ids = {"W0": 0, "b0": 1, "Wo": 2, "bo": 3, "U": 100, "V": 101}
        kw = dict(model=L.MODEL_DISGANMF, d_layers=1, d_act=str(rng.choice(["linear", "tanh", "relu"])), m=0.0)
    else:
        w = {"We": rng.randn(N, e) * 0.03, "be": rng.randn(e) * 0.01, "Wd": rng.randn(e, N) * 0.03, "bd": rng.randn(N) * 0.01, "U": rng.randn(U, k) * 0.1, "V": rng.randn(N, k) * 0.1}
        ids = {"We": 0, "be": 1, "Wd": 2, "bd": 3, "U": 100, "V": 101}
        kw = dict(m=10.0)
    w = {n: v.astype(np.float32) for n, v in w.items()}
    perms = [rng.permutation(U) for _ in range(2)]
    uids = perms[0][:B]
    what = str((mfma, "DisGANMF" if dis else "GANMF", U, N, k, e, B))

    def run(multi, tune, epochs):
        monkeypatch.setenv("GANMF_MULTI", str(multi))
        monkeypatch.setenv("GANMF_TUNE", tune)
        eng = Engine(U, N, k, e, B, mfma=mfma, **kw, **hp)
        eng.set_urm(urm)
        for n, tid in ids.items():
            eng.set_tensor(tid, w[n])
        if epochs:
            losses = [eng.train_epoch(p, 1, 1) for p in perms]
        else:
            losses = [(np.array([eng.train_step(0, uids)]), np.array([eng.train_step(1, uids)]))]
        out = {n: eng.get_tensor(tid).copy() for n, tid in ids.items()}
        out.update({n + ".m": eng.get_tensor(tid, slot=L.SLOT_ADAM_M).copy() for n, tid in ids.items()})
        eng.close()
        return out, losses
    if not dis:
        ref, ref_l = run(65, "bf16w=0", True)
        got, got_l = run(127, "bf16w=0", True)
        for n in ref:
            np.testing.assert_array_equal(got[n], ref[n], err_msg="%s %s" % (n, what))
        for (dl, gl), (dr, gr) in zip(got_l, ref_l):
            np.testing.assert_array_equal(dl, dr, err_msg=what)
            np.testing.assert_array_equal(gl, gr, err_msg=what)
    ref, ref_l = run(64, "bf16w=0,dis_top_gw=0,dis_uid_top=0", False)
    got, got_l = run(127, "", False)
    # a gradient element moves by its single-rounding error when the order of a low-precision sum changes (or the generator product is split along K)
    tol = 3e-3 if mfma == "f16" else 2e-2
    for n in ids:
        err = _err(got[n + ".m"], ref[n + ".m"].astype(np.float64))
        assert err <= tol, (n, what, err)
        lr = hp["g_lr"] if n in ("U", "V") else hp["d_lr"]
        assert _err(got[n], ref[n].astype(np.float64)) <= 2.2 * lr / np.max(np.abs(ref[n])) + 1e-6, (n, what)      # <= 2 lr per update
    np.testing.assert_allclose(got_l[0][0], ref_l[0][0], rtol=2e-3, atol=1e-6, err_msg=what)
    np.testing.assert_allclose(got_l[0][1], ref_l[0][1], rtol=2e-3, atol=1e-6, err_msg=what)
