"""Combined launches of the step (csrc/gemm_multi.hpp) against the one-kernel-per-piece path.

A combined launch runs the bodies of the ordinary kernels as block ranges of one grid (generator GEMM + CSR row expansion;
gUb + gV with the update of V written to a second buffer; the slab sum of dE inside the gWd launch; d_coef inside the dE
launch; gWd + gWe as block ranges of one grid; the gUb slabs summed by adam_rows_kernel), so it must reproduce the separate launches BIT FOR BIT: same arithmetic, same summation order.  GANMF_MULTI
(bits 0-4 the combined launches, bit 5 = 32 the gUb slabs summed by adam_rows_kernel) is read when a handle is created."""
import numpy as np
import pytest

from ganmf_amd.synthetic import glorot_params, synthetic_urm

pytestmark = pytest.mark.gpu

IDS = {"We": 0, "be": 1, "Wd": 2, "bd": 3, "U": 100, "V": 101}


def _run(monkeypatch, multi, defer, U, N, k, e, B, hp, epochs, d_steps=1, g_steps=1, mfma=None):
    from ganmf_amd import _lib as L
    from ganmf_amd.engine import Engine
    monkeypatch.setenv("GANMF_MULTI", str(multi + 32 * defer + 64))
    urm = synthetic_urm(U, N, 0.04, seed=21)
    w = glorot_params(U, N, k, e, seed=9)
    eng = Engine(U, N, k, e, B, mfma=mfma, **hp)
    eng.set_urm(urm)
    for n, tid in IDS.items():
        eng.set_tensor(tid, w[n])
    rng = np.random.RandomState(5)
    losses = []
    for _ in range(epochs):
        dl, gl = eng.train_epoch(rng.permutation(U), d_steps, g_steps)
        losses.append((np.array(dl), np.array(gl)))
    out = {n: eng.get_tensor(tid).copy() for n, tid in IDS.items()}
    out.update({n + ".m": eng.get_tensor(tid, slot=L.SLOT_ADAM_M).copy() for n, tid in IDS.items()})
    out.update({n + ".v": eng.get_tensor(tid, slot=L.SLOT_ADAM_V).copy() for n, tid in IDS.items()})
    out["scores"] = eng.scores(np.arange(min(U, 64)))
    eng.close()
    return out, losses


@pytest.mark.parametrize("shape", [
    (1500, 3706, 250, 992, 128),       # C2-shaped: every combined launch is taken (16-wave fp32 ring plans, split dE and gUb)
    (700, 1100, 64, 200, 96),          # smaller: ragged last batch, different split counts
])
@pytest.mark.parametrize("g_reg", [0.0, 1e-3])
def test_combined_launches_bit_identical(shape, g_reg, monkeypatch):
    U, N, k, e, B = shape
    hp = dict(d_lr=1e-4, g_lr=2e-4, d_reg=1e-4, g_reg=g_reg, m=10.0, recon_coefficient=0.05)
    ref, ref_l = _run(monkeypatch, 0, 0, U, N, k, e, B, hp, epochs=2)
    for multi, defer in ((31, 1), (15, 1), (1, 0), (2, 1), (4, 0), (12, 0), (7, 1), (20, 0), (28, 1)):
        got, got_l = _run(monkeypatch, multi, defer, U, N, k, e, B, hp, epochs=2)
        for (dl, gl), (dr, gr) in zip(got_l, ref_l):
            np.testing.assert_array_equal(dl, dr, err_msg="D losses, GANMF_MULTI=%d" % multi)
            np.testing.assert_array_equal(gl, gr, err_msg="G losses, GANMF_MULTI=%d" % multi)
        for n in ref:
            np.testing.assert_array_equal(got[n], ref[n], err_msg="%s, GANMF_MULTI=%d" % (n, multi + 32 * defer + 64))


@pytest.mark.parametrize("shape", [
    (700, 1100, 64, 200, 96),          # 18 tile columns of gWd over 8 XCD rectangles of 2 or 3
    (300, 300, 16, 70, 64),            # 5 tile columns: three of the eight rectangles are empty
    (900, 3706, 32, 992, 128),         # the ML-1M tile grid (16 x 58)
])
def test_wgrad_blocked_tile_order_bit_identical(shape, monkeypatch):
    """GANMF_TUNE wgrad_xb: gWd of the fused-Adam pair launch walks its tiles XCD rectangle by rectangle (default: bands of 16 tile
    rows) instead of in list order.  Every tile is the same sum and the same in-place Adam update whatever the order."""
    U, N, k, e, B = shape
    hp = dict(d_lr=1e-4, g_lr=2e-4, d_reg=1e-4, g_reg=0.0, m=10.0, recon_coefficient=0.05)
    monkeypatch.setenv("GANMF_TUNE", "wgrad_xb=0")
    ref, ref_l = _run(monkeypatch, 31, 1, U, N, k, e, B, hp, epochs=2)
    for xb in ("16", "2", "1"):
        monkeypatch.setenv("GANMF_TUNE", "wgrad_xb=" + xb)
        got, got_l = _run(monkeypatch, 31, 1, U, N, k, e, B, hp, epochs=2)
        for (dl, gl), (dr, gr) in zip(got_l, ref_l):
            np.testing.assert_array_equal(dl, dr, err_msg="D losses, wgrad_xb=" + xb)
            np.testing.assert_array_equal(gl, gr, err_msg="G losses, wgrad_xb=" + xb)
        for n in ref:
            np.testing.assert_array_equal(got[n], ref[n], err_msg="%s, wgrad_xb=%s" % (n, xb))


@pytest.mark.parametrize("mfma", ["f16", "bf16"])
@pytest.mark.parametrize("shape", [
    (1500, 3706, 250, 992, 128),       # C2-shaped: front and pair are taken on the one-piece 16-wave loop (staged discriminator passes too)
    (330, 1100, 64, 200, 96),          # three full minibatches + a ragged one: per-step front launches
])
def test_low_precision_combined_launches_bit_identical(shape, mfma, monkeypatch):
    """A handle created with mfma = "f16" | "bf16" (BASELINE configs[4] as written) takes the combined launches as well (round 6): generator GEMM +
    CSR rows (front_lp_kernel) and gUb + gV with Adam (pair_lp_kernel) run the 16-wave ONE-piece loop their products run stand-alone
    (gemm_bf16k_mfma<.., 1, F16>), so tensors, moments and losses equal the one-kernel-per-piece path bit for bit.  (bf16w = 0 on both sides: the
    unsplit 64 x 32 form of dF / decode sums in another order than two K slices + a slab sum; tests/test_gpu_gemm.py holds it to the fp64 product.)"""
    U, N, k, e, B = shape
    hp = dict(d_lr=1e-4, g_lr=2e-4, d_reg=1e-4, g_reg=0.0, m=10.0, recon_coefficient=0.05)
    monkeypatch.setenv("GANMF_TUNE", "bf16w=0")
    monkeypatch.setenv("GANMF_DEBUG_PLAN", "1")
    ref, ref_l = _run(monkeypatch, 0, 0, U, N, k, e, B, hp, epochs=2, mfma=mfma)
    for multi, defer in ((31, 1), (1, 1), (2, 1)):
        got, got_l = _run(monkeypatch, multi, defer, U, N, k, e, B, hp, epochs=2, mfma=mfma)
        for (dl, gl), (dr, gr) in zip(got_l, ref_l):
            np.testing.assert_array_equal(dl, dr, err_msg="D losses, %s GANMF_MULTI=%d" % (mfma, multi))
            np.testing.assert_array_equal(gl, gr, err_msg="G losses, %s GANMF_MULTI=%d" % (mfma, multi))
        for n in ref:
            np.testing.assert_array_equal(got[n], ref[n], err_msg="%s, %s GANMF_MULTI=%d" % (n, mfma, multi + 32 * defer + 64))


@pytest.mark.parametrize("shape,d_steps", [
    ((900, 3706, 32, 992, 128), 2),    # the ML-1M tile grid, staged discriminator passes (every step its own lr_t slot), two passes per call
    ((230, 300, 16, 70, 64), 1),       # three full minibatches: no staged pass, the lr_t slot alternates; ragged last minibatch
])
def test_wd_on_the_side_lane_bit_identical(shape, d_steps, monkeypatch):
    """GANMF_TUNE wd_lane (off by default: measured slower, profiles/r06_wd_lane.md): gWd_ext + Adam(Wd) of a discriminator step on the side
    lane, under the next step's encode GEMM, joined in front of the next reader of Wd (the next decode GEMM, the generator pass, the end of
    the call); gWe_ext + Adam(We) on the main lane.  The same kernel with an empty block range for the other product: every tensor, both
    Adam moments and every loss bit for bit what the paired launch gives.  1: forked behind gWe_ext, 2: behind the slab sum of dE."""
    U, N, k, e, B = shape
    hp = dict(d_lr=1e-4, g_lr=2e-4, d_reg=1e-4, g_reg=0.0, m=10.0, recon_coefficient=0.05)
    monkeypatch.setenv("GANMF_TUNE", "wd_lane=0")
    ref, ref_l = _run(monkeypatch, 31, 1, U, N, k, e, B, hp, epochs=2, d_steps=d_steps)
    for mode in ("1", "2"):
        monkeypatch.setenv("GANMF_TUNE", "wd_lane=" + mode)
        got, got_l = _run(monkeypatch, 31, 1, U, N, k, e, B, hp, epochs=2, d_steps=d_steps)
        for (dl, gl), (dr, gr) in zip(got_l, ref_l):
            np.testing.assert_array_equal(dl, dr, err_msg="D losses, wd_lane=" + mode)
            np.testing.assert_array_equal(gl, gr, err_msg="G losses, wd_lane=" + mode)
        for n in ref:
            np.testing.assert_array_equal(got[n], ref[n], err_msg="%s, wd_lane=%s" % (n, mode))


@pytest.mark.parametrize("shape,d_steps", [
    ((900, 3706, 32, 992, 128), 2),
    ((230, 300, 16, 70, 64), 1),
])
def test_dE_slab_sum_inside_the_weight_gradient_launch_bit_identical(shape, d_steps, monkeypatch):
    """GANMF_TUNE wgrad_seam=1: the slab sum of dE as the first block range of the fused weight-gradient launch (wgrad_seam_kernel: write-through stores,
    an agent-scope arrival counter, one acquire in every gWe workgroup) instead of a launch of its own.  The same sums by the same bodies: every tensor,
    both Adam moments and every loss bit for bit; two handles in a row (the counter is per handle and monotonic)."""
    U, N, k, e, B = shape
    hp = dict(d_lr=1e-4, g_lr=2e-4, d_reg=1e-4, g_reg=0.0, m=10.0, recon_coefficient=0.05)
    monkeypatch.setenv("GANMF_TUNE", "wgrad_seam=0")
    ref, ref_l = _run(monkeypatch, 31, 1, U, N, k, e, B, hp, epochs=3, d_steps=d_steps)
    monkeypatch.setenv("GANMF_TUNE", "wgrad_seam=1")
    for _ in range(2):
        got, got_l = _run(monkeypatch, 31, 1, U, N, k, e, B, hp, epochs=3, d_steps=d_steps)
        for (dl, gl), (dr, gr) in zip(got_l, ref_l):
            np.testing.assert_array_equal(dl, dr, err_msg="D losses, wgrad_seam=1")
            np.testing.assert_array_equal(gl, gr, err_msg="G losses, wgrad_seam=1")
        for n in ref:
            np.testing.assert_array_equal(got[n], ref[n], err_msg="%s, wgrad_seam=1" % n)


def test_second_item_buffer_survives_snapshot_and_restore(monkeypatch):
    """The fused gV update ping-pongs item_embeddings between two buffers: best-weights snapshot / restore and a tensor
    upload in the middle of training must act on the live one (an odd number of generator steps leaves it in the second)."""
    from ganmf_amd.engine import Engine
    U, N, k, e, B = 300, 3706, 250, 992, 128      # 3 generator steps per epoch
    hp = dict(d_lr=1e-4, g_lr=2e-4, d_reg=1e-4, g_reg=0.0, m=10.0, recon_coefficient=0.05)
    urm = synthetic_urm(U, N, 0.04, seed=3)
    w = glorot_params(U, N, k, e, seed=4)
    eng = Engine(U, N, k, e, B, **hp)
    eng.set_urm(urm)
    for n, tid in IDS.items():
        eng.set_tensor(tid, w[n])
    perm = np.random.RandomState(0).permutation(U)
    eng.train_epoch(perm)
    v1 = eng.get_tensor(101).copy()
    assert not np.array_equal(v1, w["V"])
    eng.snapshot_best()
    eng.train_epoch(perm)
    assert not np.array_equal(eng.get_tensor(101), v1)
    eng.restore_best()
    np.testing.assert_array_equal(eng.get_tensor(101), v1)
    s1 = eng.scores(np.arange(16))
    np.testing.assert_allclose(s1, eng.get_tensor(100)[:16] @ v1.T, rtol=2e-5, atol=1e-6)
    eng.set_tensor(101, w["V"])                  # upload into the live buffer
    np.testing.assert_array_equal(eng.get_tensor(101), w["V"])
    eng.train_epoch(perm)                        # and the next update starts from it
    assert np.max(np.abs(eng.get_tensor(101) - w["V"])) <= 3 * 2.1 * hp["g_lr"]      # |delta| <= ~lr per Adam step
    eng.close()


def _run_staged(monkeypatch, tune, model, U, N, k, e, B, hp, epochs, d_steps, g_steps=1):
    """Epochs with / without the per-pass forms (GANMF_TUNE=pass_stage=0|1,lazy_rows=0|1)."""
    from ganmf_amd import _lib as L
    from ganmf_amd.engine import Engine
    monkeypatch.setenv("GANMF_TUNE", tune)
    urm = synthetic_urm(U, N, 0.04, seed=21)
    rng = np.random.RandomState(11)
    if model == "ganmf":
        ids = dict(IDS)
        w = glorot_params(U, N, k, e, seed=9)
        eng = Engine(U, N, k, e, B, **hp)
    else:
        layers = 2
        ids = {"W0": 0, "b0": 1, "W1": 2, "b1": 3, "Wo": 4, "bo": 5, "U": 100, "V": 101}
        w = {"W0": rng.randn(N + 1, e) * 0.03, "b0": rng.randn(e) * 0.01, "W1": rng.randn(e, e) * 0.05, "b1": rng.randn(e) * 0.01,
             "Wo": rng.randn(e, 1) * 0.1, "bo": np.zeros(1), "U": rng.randn(U, k) * 0.1, "V": rng.randn(N, k) * 0.1}
        w["W0"][0, :] *= 1.0 / U      # the float(uid) row
        w = {n: v.astype(np.float32) for n, v in w.items()}
        eng = Engine(U, N, k, e, B, model=L.MODEL_DISGANMF, d_layers=layers, d_act="tanh", m=0.0, **hp)
    eng.set_urm(urm)
    for n, tid in ids.items():
        eng.set_tensor(tid, w[n])
    prng = np.random.RandomState(5)
    losses = []
    for ep in range(epochs):
        perm = prng.permutation(U)
        if ep % 2 == 1:
            perm = perm[:U - U // 3]      # a call that leaves a third of the rows out: they still take every all-rows update
        dl, gl = eng.train_epoch(perm, d_steps, g_steps)
        losses.append((np.array(dl), np.array(gl)))
    out = {n: eng.get_tensor(tid).copy() for n, tid in ids.items()}
    out.update({n + ".m": eng.get_tensor(tid, slot=L.SLOT_ADAM_M).copy() for n, tid in ids.items()})
    out.update({n + ".v": eng.get_tensor(tid, slot=L.SLOT_ADAM_V).copy() for n, tid in ids.items()})
    out["scores"] = eng.scores(np.arange(min(U, 64)))
    eng.close()
    return out, losses


@pytest.mark.parametrize("case", [
    ("ganmf", 1500, 3706, 250, 992, 128, 1, 1, 0.0),     # the row expansion rides in the generator launch (front_kernel); ragged last minibatch; gUb + gV paired
    ("ganmf", 700, 1100, 20, 64, 32, 2, 3, 0.0),         # stand-alone row expansion, generator product, gUb, gV; two D and three G passes per call
    ("ganmf", 700, 1100, 20, 64, 32, 1, 1, 1e-3),        # g_reg != 0: the all-rows update stays per step (every row has a gradient every step)
    ("disganmf", 900, 1100, 64, 128, 64, 1, 2, 0.0),     # float(uid) column in the staged rows
    ("ganmf", 1500, 3706, 250, 992, 128, 0, 2, 0.0),     # generator passes only: the CSR rows alone are staged in front of the call (round 6)
    ("disganmf", 900, 1100, 64, 128, 64, 0, 1, 0.0),
])
def test_per_pass_forms_are_bit_identical(case, monkeypatch):
    """Work that a pass's frozen half of the model makes independent of the steps before it, done once per pass:
    * stage_pass (lib/step_ganmf.inc): the CSR rows and the generated rows of every full minibatch of a DISCRIMINATOR pass (U, V frozen)
      formed in front of it -- same tiles, same K order, lr_t from open_steps_kernel;
    * lazy_pass_begin / lazy_pass_end: the all-rows Adam over user_embeddings of a GENERATOR pass (g_reg == 0: a row is read once and has
      a gradient once per pass) as one advance launch in front of the pass and one flush launch behind it;
    * (round 6) the generator passes read their real rows from the staged blocks -- the reference walks the same slices of one shuffle in both
      loops (GANMF.py:175-203) -- so a generator step launches no row expansion (g_rows_staged = 2: wherever possible; the default 1: only where that expansion is a launch of its own, i.e. not inside front_kernel; 0: never);
    against every step doing its own (GANMF_TUNE=pass_stage=0,lazy_rows=0)."""
    model, U, N, k, e, B, d_steps, g_steps, g_reg = case
    hp = dict(d_lr=1e-4, g_lr=2e-4, d_reg=1e-4, g_reg=g_reg, recon_coefficient=0.05)
    if model == "ganmf":
        hp.update(m=10.0)
    ref, ref_l = _run_staged(monkeypatch, "pass_stage=0,lazy_rows=0", model, U, N, k, e, B, hp, 2, d_steps, g_steps)
    for tune in ("pass_stage=1,lazy_rows=0", "pass_stage=0,lazy_rows=1", "pass_stage=1,lazy_rows=1", "pass_stage=1,lazy_rows=1,g_rows_staged=0",
                 "pass_stage=1,lazy_rows=1,g_rows_staged=2"):
        got, got_l = _run_staged(monkeypatch, tune, model, U, N, k, e, B, hp, 2, d_steps, g_steps)
        for (dl, gl), (dr, gr) in zip(got_l, ref_l):
            np.testing.assert_array_equal(dl, dr, err_msg="D losses, " + tune)
            np.testing.assert_array_equal(gl, gr, err_msg="G losses, " + tune)
        assert set(ref) == set(got)
        for n in ref:
            np.testing.assert_array_equal(got[n], ref[n], err_msg="%s, %s" % (n, tune))


def test_lazy_pass_arena_pads_are_zero_whatever_the_allocator_returns(monkeypatch):
    """The per-pass gUb arena (lib/step_ganmf.inc lazy_chunk_alloc) comes from plain hipMalloc; GEMM and slab stores mask the columns
    >= k, and adam_rows_flush_kernel applies the update to whole ldk-wide rows.  A recycled allocation holding a NaN bit pattern in a
    pad column would therefore reach the pads of user_embeddings and, through F = Ub . V^T, every score.  The test fills device
    memory of the arena's size classes with 0xFF bytes (a NaN pattern), frees it, and then runs lazy generator passes with k far from
    a multiple of 64: scores must stay finite and equal the per-step path bit for bit."""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
    hip.hipFree.argtypes = [C.c_void_p]

    def poison():
        blocks = []
        for size in [64 << 20] * 6 + [512 << 20, 128 << 20, 32 << 20, 16 << 20]:
            p = C.c_void_p()
            if hip.hipMalloc(C.byref(p), size) != 0:
                break
            assert hip.hipMemset(p, 0xFF, size) == 0
            blocks.append(p)
        assert hip.hipDeviceSynchronize() == 0
        for p in blocks:
            assert hip.hipFree(p) == 0
        assert len(blocks) >= 6

    U, N, k, e, B = 700, 1100, 20, 64, 32
    hp = dict(d_lr=1e-4, g_lr=2e-4, d_reg=1e-4, g_reg=0.0, m=10.0, recon_coefficient=0.05)
    ref, ref_l = _run_staged(monkeypatch, "pass_stage=0,lazy_rows=0", "ganmf", U, N, k, e, B, hp, 3, 1, 2)
    poison()
    got, got_l = _run_staged(monkeypatch, "pass_stage=1,lazy_rows=1", "ganmf", U, N, k, e, B, hp, 3, 1, 2)
    assert np.all(np.isfinite(got["scores"]))
    for n in ref:
        np.testing.assert_array_equal(got[n], ref[n], err_msg=n)
    for (dl, gl), (dr, gr) in zip(got_l, ref_l):
        np.testing.assert_array_equal(dl, dr)
        np.testing.assert_array_equal(gl, gr)
