"""The data-parallel decomposition (ganmf_amd/dist.py + the per-rank arithmetic of the C++
d_step/g_step) checked on CPU with torch.distributed gloo, world_size 2: each rank computes the
oracle's gradients on its own rows with the GLOBAL batch size in the loss scales, gradients of
replicated tensors are all-reduced (or reduce-scattered, updated slice-wise and all-gathered), and the
result of a discriminator pass AND a generator pass -- item_embeddings' slice update, the rank-owned rows of
user_embeddings -- must equal the single-process oracle on the union batches."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_bounds_and_plan():
    from ganmf_amd.dist import epoch_plan, shard_bounds
    b = shard_bounds(10, 4)
    assert b == [(0, 3), (3, 6), (6, 8), (8, 10)]
    steps, rows = epoch_plan([7, 3], 4)
    assert steps == 2 and rows.tolist() == [7, 3]      # step 0: 4+3, step 1: 3+0 (rank 1 out of rows)
    steps, rows = epoch_plan([6040] * 8, 128)
    assert steps == 48 and rows[0] == 1024 and rows[-1] == 8 * (6040 - 47 * 128)


def _worker(rank, world, port, out, sliced=False):
    import torch.distributed as dist
    import torch
    sys.path.insert(0, ROOT)
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    from oracle.ganmf_oracle import GANMFOracle
    from ganmf_amd.dist import epoch_plan, shard_bounds
    rng = np.random.RandomState(0)
    U, N, k, e, B = 23, 31, 4, 6, 8
    X = (rng.rand(U, N) < 0.2).astype(np.float64)
    hp = dict(d_lr=1e-3, g_lr=1e-3, d_reg=1e-3, g_reg=1e-3, m=10.0, recon_coefficient=0.1)
    full = GANMFOracle(U, N, k, e, dtype=np.float64, seed=4, **hp)
    bounds = shard_bounds(U, world)
    lo, hi = bounds[rank]
    steps, grows = epoch_plan([b - a for a, b in bounds], B)
    # local model: own rows of U, replicated everything else
    loc = GANMFOracle(hi - lo, N, k, e, dtype=np.float64, seed=4, **hp)
    p = full.get_params()
    loc.set_params(We=p["We"], be=p["be"], Wd=p["Wd"], bd=p["bd"], V=p["V"], U=p["U"][lo:hi])
    for i in range(steps):
        rows = np.arange(i * B, min((i + 1) * B, hi - lo))
        Bg = int(grows[i])
        # ---- D step with global scales: sums all-reduced before the hinge
        nb = len(rows)
        if nb:
            F = loc.generator(rows)
            Er, dr, _ = loc.autoencoder(X[lo:hi][rows]); Ef, df, _ = loc.autoencoder(F)
            sums = torch.tensor([np.sum(dr * dr), np.sum(df * df)])
        else:
            sums = torch.zeros(2, dtype=torch.float64)
        dist.all_reduce(sums)
        Lr, Lf = sums[0].item() / (Bg * N), sums[1].item() / (Bg * N)
        on = (hp["m"] * Lr - Lf) > 0
        g = {n: np.zeros_like(loc.p[n]) for n in loc.D_NAMES}
        if nb:
            s = 2.0 / (Bg * N)
            for inp, E, dl, c in ((X[lo:hi][rows], Er, dr, 1 + (hp["m"] if on else 0)), (F, Ef, df, -1.0 if on else 0.0)):
                dR = c * s * dl
                g["Wd"] += E.T @ dR; g["bd"] += dR.sum(0)
                dE = dR @ loc.p["Wd"].T
                g["We"] += inp.T @ dE; g["be"] += dE.sum(0)
        for n in loc.D_NAMES:
            if not sliced:
                t = torch.from_numpy(g[n]); dist.all_reduce(t)
                g[n] = t.numpy() + hp["d_reg"] * loc.p[n]
                loc.opt_d.apply_dense(n, loc.p[n], g[n])
                continue
            # the library's form (csrc/lib/dataparallel.inc dp_update): the summed gradient is only needed on the owner of each slice
            # (reduce-scatter; gloo has none, an all-reduce of which the rank keeps its slice stands in), TF-Adam on that
            # slice of the flattened, zero-padded parameter with moments that exist on the owner alone, then an
            # all-gather of the parameter slices
            flat = loc.p[n].reshape(-1)
            slice_n = -(-flat.size // world)
            gpad = np.zeros(slice_n * world); gpad[:flat.size] = g[n].reshape(-1)
            ppad = np.zeros(slice_n * world); ppad[:flat.size] = flat
            t = torch.from_numpy(gpad); dist.all_reduce(t)
            sl = slice(rank * slice_n, (rank + 1) * slice_n)
            mine = ppad[sl].copy()
            loc.opt_d.apply_dense(n, mine, t.numpy()[sl] + hp["d_reg"] * mine)
            parts = [torch.zeros(slice_n, dtype=torch.float64) for _ in range(world)]
            dist.all_gather(parts, torch.from_numpy(mine))
            loc.p[n][...] = torch.cat(parts).numpy()[:flat.size].reshape(loc.p[n].shape)
        loc.opt_d.finish()
        # reference: one oracle D-step on the union batch
        union = np.concatenate([np.arange(a + i * B, min(a + (i + 1) * B, b)) for a, b in bounds])
        full.d_step(union, X[union])
    for n in loc.D_NAMES:
        np.testing.assert_allclose(loc.p[n], full.p[n], rtol=1e-9, atol=1e-12)
    # ---- the generator pass over the SAME slices (GANMF.py:191-203; csrc/lib/step_ganmf.inc g_step / gen_update): every rank forms
    # dF for its own rows with the GLOBAL batch size in both scales; gV is summed over ranks (all-reduce, or reduce-scatter
    # -> Adam on the rank's slice of V -> all-gather); the rows of U belong to their rank and are never communicated -- the
    # all-rows TF update (App. B.5: non-batch rows decay their moments and still move) runs on the local rows only.
    alpha, g_reg = hp["recon_coefficient"], hp["g_reg"]
    for i in range(steps):
        rows = np.arange(i * B, min((i + 1) * B, hi - lo))
        Bg, nb = int(grows[i]), len(rows)
        gV = np.zeros_like(loc.p["V"])
        gU = g_reg * loc.p["U"]
        if nb:
            Xb = X[lo:hi][rows]
            Ub = loc.p["U"][rows]
            F = Ub @ loc.p["V"].T
            Er = Xb @ loc.p["We"] + loc.p["be"]
            Ef, df, _ = loc.autoencoder(F)
            dR = ((1 - alpha) * 2.0 / (Bg * N)) * df
            dE = dR @ loc.p["Wd"].T + (alpha * 2.0 / (Bg * e)) * (Ef - Er)
            dF = dE @ loc.p["We"].T - dR
            gV = dF.T @ Ub
            gU[rows] += dF @ loc.p["V"]           # reads the OLD V
        if not sliced:
            t = torch.from_numpy(gV); dist.all_reduce(t)
            loc.opt_g.apply_dense("V", loc.p["V"], t.numpy() + g_reg * loc.p["V"])
        else:
            flat = loc.p["V"].reshape(-1)
            slice_n = -(-flat.size // world)
            gpad = np.zeros(slice_n * world); gpad[:flat.size] = gV.reshape(-1)
            ppad = np.zeros(slice_n * world); ppad[:flat.size] = flat
            t = torch.from_numpy(gpad); dist.all_reduce(t)
            sl = slice(rank * slice_n, (rank + 1) * slice_n)
            mine = ppad[sl].copy()
            loc.opt_g.apply_dense("V", mine, t.numpy()[sl] + g_reg * mine)
            parts = [torch.zeros(slice_n, dtype=torch.float64) for _ in range(world)]
            dist.all_gather(parts, torch.from_numpy(mine))
            loc.p["V"][...] = torch.cat(parts).numpy()[:flat.size].reshape(loc.p["V"].shape)
        loc.opt_g.apply_sparse_all_rows("U", loc.p["U"], gU)
        loc.opt_g.finish()
        union = np.concatenate([np.arange(a + i * B, min(a + (i + 1) * B, b)) for a, b in bounds])
        full.g_step(union, X[union])
    np.testing.assert_allclose(loc.p["V"], full.p["V"], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(loc.p["U"], full.p["U"][lo:hi], rtol=1e-9, atol=1e-12)      # rank-owned rows
    dist.barrier()
    dist.destroy_process_group()
    out.put((rank, "ok"))


@pytest.mark.parametrize("sliced", [False, True])
def test_sharded_d_and_g_steps_equal_union_batch_gloo(sliced):
    """sliced = False: all-reduce + replicated Adam; True: reduce-scatter / Adam on the rank's slice / all-gather (what the
    library runs since round 2).  Both must equal the single-process oracle on the union batches."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + (7 if sliced else 0)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, sliced)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    got = sorted(q.get(timeout=5) for _ in range(2))
    assert got == [(0, "ok"), (1, "ok")]
