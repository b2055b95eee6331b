"""The oracle's hand-written backward formulas vs torch.autograd applied to the literal loss
expressions of the reference (GANMF.py:131-135, DisGANMF.py:114-136).  CPU only, fp64."""
import numpy as np
import pytest
import torch

from oracle.ganmf_oracle import ACTIVATIONS, DisGANMFOracle, GANMFOracle


def _t(a, grad=False):
    return torch.tensor(np.asarray(a, dtype=np.float64), requires_grad=grad)


def _rand_batch(rng, U, N, B, density=0.2):
    uids = rng.permutation(U)[:B]
    X = (rng.rand(B, N) < density).astype(np.float64)
    return uids, X


@pytest.mark.parametrize("m,scale_fake", [(10.0, 1.0), (0.01, 5.0)])  # hinge active / inactive
def test_ganmf_d_and_g_grads(m, scale_fake):
    rng = np.random.RandomState(0)
    U, N, k, e, B = 23, 31, 5, 7, 6
    o = GANMFOracle(U, N, k, e, d_reg=1e-2, g_reg=3e-3, m=m, recon_coefficient=0.3, dtype=np.float64, seed=3)
    o.set_params(be=rng.randn(e) * 0.1, bd=rng.randn(N) * 0.1, U=o.p["U"] * scale_fake)
    uids, X = _rand_batch(rng, U, N, B)

    P = {n: _t(v, True) for n, v in o.p.items()}
    Xt = _t(X)

    def ae(inp):
        E = inp @ P["We"] + P["be"]
        R = E @ P["Wd"] + P["bd"]
        return E, torch.mean((R - inp) ** 2)

    F = P["U"][torch.tensor(uids)] @ P["V"].T
    Er, Lr = ae(Xt)
    Ef, Lf = ae(F)
    l2 = lambda names: sum((P[n] ** 2).sum() / 2 for n in names)
    dloss = Lr + torch.clamp(m * Lr - Lf, min=0.0) + 1e-2 * l2(["We", "be", "Wd", "bd"])
    gloss = (1 - 0.3) * Lf + 0.3 * torch.mean((Er - Ef) ** 2) + 3e-3 * l2(["U", "V"])
    gd = torch.autograd.grad(dloss, [P[n] for n in o.D_NAMES], retain_graph=True)
    gg = torch.autograd.grad(gloss, [P[n] for n in o.G_NAMES])

    loss_d, g_d = o.d_grads(uids, X)
    loss_g, g_g = o.g_grads(uids, X)
    assert o.hinge_active_last == (m > 1)
    assert abs(loss_d - dloss.item()) < 1e-12
    assert abs(loss_g - gloss.item()) < 1e-12
    for n, t in zip(o.D_NAMES, gd):
        np.testing.assert_allclose(g_d[n], t.numpy(), rtol=1e-10, atol=1e-13)
    for n, t in zip(o.G_NAMES, gg):
        np.testing.assert_allclose(g_g[n], t.numpy(), rtol=1e-10, atol=1e-13)


@pytest.mark.parametrize("act", ACTIVATIONS)
@pytest.mark.parametrize("layers", [1, 2])
def test_disganmf_grads(act, layers):
    rng = np.random.RandomState(1)
    U, N, k, e, B = 19, 17, 4, 6, 5
    o = DisGANMFOracle(U, N, k, d_layers=layers, d_nodes=e, d_hidden_act=act, d_reg=2e-2, g_reg=1e-3,
                       recon_coefficient=0.4, dtype=np.float64, seed=5)
    uids, X = _rand_batch(rng, U, N, B)
    P = {n: _t(v, True) for n, v in o.p.items()}
    tact = {"linear": lambda z: z, "tanh": torch.tanh, "relu": torch.relu, "sigmoid": torch.sigmoid}[act]

    def disc(inp):
        h = torch.cat([_t(uids).reshape(-1, 1), inp], dim=1)
        for l in range(layers):
            h = tact(h @ P["W%d" % l] + P["b%d" % l])
        return h, h @ P["Wo"] + P["bo"]

    bce = torch.nn.functional.binary_cross_entropy_with_logits
    F = P["U"][torch.tensor(uids)] @ P["V"].T
    fr, outr = disc(_t(X))
    ff, outf = disc(F)
    loss_real = bce(outr, torch.ones_like(outr))
    loss_fake = bce(outf, torch.zeros_like(outf))
    l2 = lambda names: sum((P[n] ** 2).sum() / 2 for n in names)
    dloss = loss_real + loss_fake + 2e-2 * l2(o.D_NAMES)
    gloss = loss_fake + 0.4 * torch.mean((fr - ff) ** 2) + 1e-3 * l2(o.G_NAMES)
    gd = torch.autograd.grad(dloss, [P[n] for n in o.D_NAMES], retain_graph=True)
    gg = torch.autograd.grad(gloss, [P[n] for n in o.G_NAMES])
    loss_d, g_d = o.d_grads(uids, X)
    loss_g, g_g = o.g_grads(uids, X)
    assert abs(loss_d - dloss.item()) < 1e-10
    assert abs(loss_g - gloss.item()) < 1e-10
    for n, t in zip(o.D_NAMES, gd):
        np.testing.assert_allclose(g_d[n], t.numpy(), rtol=1e-9, atol=1e-11)
    for n, t in zip(o.G_NAMES, gg):
        np.testing.assert_allclose(g_g[n], t.numpy(), rtol=1e-9, atol=1e-11)


def test_adam_matches_tf_formula_scalar():
    """ApplyAdam closed form on a scalar, 3 steps, against a hand evaluation."""
    from oracle.ganmf_oracle import _Adam
    opt = _Adam(1e-3, np.float64)
    var = np.array([1.0])
    m = v = 0.0
    b1p, b2p = 0.9, 0.999
    ref = 1.0
    for g in (0.5, -0.25, 0.125):
        opt.apply_dense("x", var, np.array([g]))
        opt.finish()
        a = 1e-3 * np.sqrt(1 - b2p) / (1 - b1p)
        m = m + (g - m) * 0.1
        v = v + (g * g - v) * 0.001
        ref -= m * a / (np.sqrt(v) + 1e-8)
        b1p *= 0.9
        b2p *= 0.999
        assert abs(var[0] - ref) < 1e-15
