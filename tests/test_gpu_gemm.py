"""fp32 MFMA GEMM kernel vs numpy fp64 through the C ABI (ganmf_gemm_f32): every operand layout,
both tile sizes, split-K, ragged edges.  Tolerance: the MFMA is an exact fp32 fma chain, so the
error bound is the fp32 summation bound ~ K * eps * sum|a||b|; we use 2e-6 * K-scaled bound."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

LAYOUTS = [(False, False), (False, True), (True, True)]  # NT, NN, TN


def _mk(rng, M, N, K, akm, bkm):
    A = rng.standard_normal((K, M) if akm else (M, K)).astype(np.float32)
    B = rng.standard_normal((K, N) if bkm else (N, K)).astype(np.float32)
    Am = A.T if akm else A
    Bm = B if bkm else B.T
    ref = Am.astype(np.float64) @ Bm.astype(np.float64)
    bound = np.abs(Am).astype(np.float64) @ np.abs(Bm).astype(np.float64)
    return A, B, ref, bound


@pytest.mark.parametrize("akm,bkm", LAYOUTS)
@pytest.mark.parametrize("tile", [64, 128])
@pytest.mark.parametrize("shape", [(128, 128, 32), (1, 1, 1), (37, 53, 5), (130, 70, 250), (256, 992, 515),
                                   (200, 3706, 96), (64, 250, 3706)])
def test_gemm_layouts(akm, bkm, tile, shape):
    from ganmf_amd.engine import gemm_f32
    M, N, K = shape
    rng = np.random.RandomState(M * 7 + N * 3 + K)
    A, B, ref, bound = _mk(rng, M, N, K, akm, bkm)
    out, _ = gemm_f32(A, B, akm, bkm, tile=tile)
    err = np.abs(out - ref)
    assert np.all(err <= 4e-7 * bound * np.sqrt(K) + 1e-30), float((err / (bound + 1e-30)).max())


@pytest.mark.parametrize("akm,bkm", LAYOUTS)
@pytest.mark.parametrize("nsplit", [2, 5, 16])
def test_gemm_splitk(akm, bkm, nsplit):
    from ganmf_amd.engine import gemm_f32
    M, N, K = 256, 250, 3706
    rng = np.random.RandomState(nsplit)
    A, B, ref, bound = _mk(rng, M, N, K, akm, bkm)
    out, _ = gemm_f32(A, B, akm, bkm, tile=0, nsplit=nsplit)
    err = np.abs(out - ref)
    assert np.all(err <= 4e-7 * bound * np.sqrt(K))


def test_gemm_identity_asymmetric():
    """A = I with an asymmetric B catches a transposed C write (cdna guide §3)."""
    from ganmf_amd.engine import gemm_f32
    n = 96
    B = (np.arange(n)[:, None] * 1000 + np.arange(n)[None, :]).astype(np.float32)  # [N, K], asymmetric
    out, _ = gemm_f32(np.eye(n, dtype=np.float32), B, False, False)
    np.testing.assert_array_equal(out, B.T)
    out, _ = gemm_f32(np.eye(n, dtype=np.float32), B, True, True)   # A^T = I, B as [K, N]
    np.testing.assert_array_equal(out, B)


def test_gemm_deterministic():
    from ganmf_amd.engine import gemm_f32
    rng = np.random.RandomState(0)
    A = rng.standard_normal((256, 3706)).astype(np.float32)
    B = rng.standard_normal((3706, 992)).astype(np.float32)
    o1, _ = gemm_f32(A, B, False, True, nsplit=8)
    o2, _ = gemm_f32(A, B, False, True, nsplit=8)
    np.testing.assert_array_equal(o1, o2)


@pytest.mark.parametrize("shape", [(128, 128, 32), (1, 1, 1), (700, 900, 250), (1300, 1500, 70), (3000, 260, 33),
                                   (129, 4100, 250), (5000, 131, 8)])
def test_gemm_persistent_tile_walk(shape, monkeypatch):
    """gemm_persist.hpp (one workgroup per CU walks a list of output tiles, stores deferred under the next tile's
    K loop) against fp64 and, bit for bit, against the one-tile-per-workgroup kernel: same K order, same MFMA chain."""
    from ganmf_amd.engine import gemm_f32
    M, N, K = shape
    rng = np.random.RandomState(M + 3 * N + 7 * K)
    A, B, ref, bound = _mk(rng, M, N, K, False, False)
    monkeypatch.setenv("GANMF_MFMA", "f32")
    monkeypatch.setenv("GANMF_TUNE", "persist=1")      # two 4-wave workgroups per CU (the one-workgroup forms are experiment builds)
    out, _ = gemm_f32(A, B, False, False, tile=128)
    err = np.abs(out - ref)
    assert np.all(err <= 4e-7 * bound * np.sqrt(K) + 1e-30), float((err / (bound + 1e-30)).max())
    monkeypatch.setenv("GANMF_TUNE", "persist=0")
    plain, _ = gemm_f32(A, B, False, False, tile=128)
    np.testing.assert_array_equal(out, plain)


@pytest.mark.parametrize("akm,bkm", LAYOUTS)
@pytest.mark.parametrize("kg", ["1", "2", "4"])
@pytest.mark.parametrize("ring", ["2", "3"])
@pytest.mark.parametrize("shape,nsplit", [((256, 992, 3707), 4), ((128, 250, 3706), 0), ((65, 70, 130), 1), ((1, 1, 1), 1),
                                           ((200, 3706, 96), 1), ((130, 129, 64), 1)])
def test_gemm_k_groups(akm, bkm, kg, ring, shape, nsplit, monkeypatch):
    """64 x 64 fp32 ring kernel with KG groups of four waves per workgroup (GANMF_TUNE kg=; the groups split the 8-wide chunks of
    every K-tile and their partial tiles meet through LDS in the epilogue): every layout, both ring depths, ragged edges,
    K shorter than one K-tile, split-K on top.  Same error bound as the one-group kernel, and run-to-run identical."""
    from ganmf_amd.engine import gemm_f32
    M, N, K = shape
    rng = np.random.RandomState(M + 3 * N + 7 * K)
    A, B, ref, bound = _mk(rng, M, N, K, akm, bkm)
    monkeypatch.setenv("GANMF_MFMA", "f32")
    monkeypatch.setenv("GANMF_TUNE", "kg=%s,ring=%s" % (kg, ring))
    out, _ = gemm_f32(A, B, akm, bkm, tile=64, nsplit=nsplit)
    err = np.abs(out - ref)
    assert np.all(err <= 4e-7 * bound * np.sqrt(K) + 1e-30), float((err / (bound + 1e-30)).max())
    again, _ = gemm_f32(A, B, akm, bkm, tile=64, nsplit=nsplit)
    np.testing.assert_array_equal(out, again)


@pytest.mark.parametrize("mfma", ["f32", "bf16x3"])
@pytest.mark.parametrize("tile", [64, 128])
@pytest.mark.parametrize("shape", [(1, 1, 1), (700, 900, 250), (129, 4100, 40), (5000, 131, 8), (1300, 1500, 70), (64 * 9, 64 * 7, 3000)])
def test_gemm_blocked_tile_order(mfma, tile, shape, monkeypatch):
    """GANMF_TUNE tile_order=2 forces the XCD-blocked block -> tile map (gemm_f32.hpp tile_coords: eight rectangles of the tile
    grid, banded, M-innermost) on every unsplit product; the map must be a bijection for ragged tile grids (fewer than 8 tiles,
    rectangles of unequal size, bands taller than a rectangle), so the result equals the list-order launch bit for bit."""
    from ganmf_amd.engine import gemm_f32
    M, N, K = shape
    rng = np.random.RandomState(M + 3 * N + 7 * K)
    A, B, ref, bound = _mk(rng, M, N, K, False, False)
    monkeypatch.setenv("GANMF_MFMA", mfma)
    monkeypatch.setenv("GANMF_TUNE", "persist=0,tile_order=0")
    plain, _ = gemm_f32(A, B, False, False, tile=tile, nsplit=1)
    monkeypatch.setenv("GANMF_TUNE", "persist=0,tile_order=2")
    out, _ = gemm_f32(A, B, False, False, tile=tile, nsplit=1)
    np.testing.assert_array_equal(out, plain)
    err = np.abs(out - ref)
    assert np.all(err <= 4e-7 * bound * np.sqrt(K) + 1e-30), float((err / (bound + 1e-30)).max())


@pytest.mark.parametrize("akm,bkm", LAYOUTS)
@pytest.mark.parametrize("shape,nsplit", [((64, 32, 17633), 0), ((64, 32, 17633), 250), ((33, 10, 9000), 100), ((200, 130, 5000), 20),
                                           ((256, 250, 3706), 29)])
def test_gemm_deep_split_small_output(akm, bkm, shape, nsplit):
    """Deep split-K behind a small output (LastFM at the reference's defaults: 64 x 32, K = 17 632): the slab sum runs with
    16 threads per output element (gemm_f32.hpp reduce_groups), partial sums added in group order -- fp32 bound and run-to-run
    identical, ragged column tails included."""
    from ganmf_amd.engine import gemm_f32
    M, N, K = shape
    rng = np.random.RandomState(M + 3 * N + 7 * K + nsplit)
    A, B, ref, bound = _mk(rng, M, N, K, akm, bkm)
    out, _ = gemm_f32(A, B, akm, bkm, tile=64, nsplit=nsplit)
    err = np.abs(out - ref)
    assert np.all(err <= 4e-7 * bound * np.sqrt(K) + 1e-30), float((err / (bound + 1e-30)).max())
    again, _ = gemm_f32(A, B, akm, bkm, tile=64, nsplit=nsplit)
    np.testing.assert_array_equal(out, again)


@pytest.mark.parametrize("akm,bkm", LAYOUTS)
@pytest.mark.parametrize("shape,nsplit", [((256, 992, 3707), 4), ((128, 992, 3706), 8), ((128, 3706, 993), 2), ((16, 7, 54), 1), ((1, 1, 1), 1),
                                           ((64, 64, 64), 1), ((65, 70, 130), 1), ((64, 64, 192), 1), ((130, 129, 256), 1), ((200, 100, 321), 1),
                                           ((200, 3706, 96), 1), ((70, 65, 1000), 3)])
def test_gemm_split_bf16_k_groups(akm, bkm, shape, nsplit, monkeypatch):
    """gemm_bf16k.hpp: the 16-wave split-bf16 kernel that takes the plans of the 16-wave fp32 ring kernel (GANMF_X3KG) -- every layout,
    K ranges of 1 .. 58 K-tiles (fewer tiles than the four prefetch slots, tile counts that are no multiple of four, K tails, a
    short last K slice), ragged tile edges, split-K.  fp32-accurate like the staged split-bf16 kernel, and run-to-run identical."""
    from ganmf_amd.engine import gemm_f32
    M, N, K = shape
    rng = np.random.RandomState(M + 3 * N + 7 * K)
    A, B, ref, bound = _mk(rng, M, N, K, akm, bkm)
    monkeypatch.setenv("GANMF_MFMA", "f32")
    monkeypatch.setenv("GANMF_TUNE", "kg=4,ring=3")
    monkeypatch.setenv("GANMF_X3KG", "3")
    out, _ = gemm_f32(A, B, akm, bkm, tile=64, nsplit=nsplit)
    err = np.abs(out - ref)
    assert np.all(err <= 4e-7 * bound * np.sqrt(K) + 1e-30), float((err / (bound + 1e-30)).max())
    again, _ = gemm_f32(A, B, akm, bkm, tile=64, nsplit=nsplit)
    np.testing.assert_array_equal(out, again)
    monkeypatch.setenv("GANMF_X3KG", "0")
    plain, _ = gemm_f32(A, B, akm, bkm, tile=64, nsplit=nsplit)
    assert not np.array_equal(plain, out) or K == 1      # (it IS another kernel)


@pytest.mark.parametrize("shape", [(128, 3706, 992), (128, 3706, 993), (64, 32, 128), (1, 1, 1), (65, 33, 129), (100, 70, 1030), (128, 2113, 748),
                                   (37, 500, 384), (200, 100, 257), (64, 96, 2000), (70, 40, 100), (33, 64, 256)])
@pytest.mark.parametrize("bkm", [False, True])
def test_gemm_bf16w_64x32_tiles(shape, bkm, monkeypatch, capfd):
    """gemm_bf16w.hpp: the 16-wave split-bf16 loop on 64 x 32 tiles with 128-deep K-tiles (eight K groups), unsplit, B K-contiguous (dF of
    the generator step) or K-major (its decode: [k][32 n] image, transposing fragment reads) -- the ML-1M and hetrec shapes, 1 .. 16 K-tiles (BF16W_PD = 2 K-tiles in flight: one K-tile, K <= 128, leaves the
    second prologue slot past the range -- requested from the zero page --, two K-tiles, K = 129 .. 256, fill both; odd and even tile counts), K tails, a last K-tile that reaches past the leading dimension (K = 1030: ld 1088 < 1152), ragged rows and
    columns.  fp32-accurate like the 64 x 64 kernel, run-to-run identical, and another kernel than the default plan's."""
    from ganmf_amd.engine import gemm_f32
    M, N, K = shape
    rng = np.random.RandomState(M + 3 * N + 7 * K)
    A, B, ref, bound = _mk(rng, M, N, K, False, bkm)
    monkeypatch.setenv("GANMF_TUNE", "bf16w=2")
    monkeypatch.setenv("GANMF_DEBUG_PLAN", "1")
    capfd.readouterr()
    out, _ = gemm_f32(A, B, False, bkm)
    assert "64 x 32 tiles" in capfd.readouterr().err
    err = np.abs(out - ref)
    assert np.all(err <= 4e-7 * bound * np.sqrt(K) + 1e-30), float((err / (bound + 1e-30)).max())
    again, _ = gemm_f32(A, B, False, bkm)
    np.testing.assert_array_equal(out, again)
    monkeypatch.setenv("GANMF_TUNE", "bf16w=0")
    plain, _ = gemm_f32(A, B, False, bkm)
    assert np.all(np.abs(plain - out) <= 8e-7 * bound * np.sqrt(K) + 1e-30)
    assert not np.array_equal(plain, out) or K == 1


@pytest.mark.parametrize("mode,bound", [("f16", 1.5e-3), ("bf16", 1.2e-2)])
@pytest.mark.parametrize("bkm", [False, True])
@pytest.mark.parametrize("shape", [(128, 3706, 1024), (65, 33, 129), (100, 70, 1030), (37, 500, 100), (64, 96, 2000)])
def test_gemm_bf16w_one_piece_forms(shape, bkm, mode, bound, monkeypatch, capfd):
    """gemm_bf16w.hpp with ONE low-precision piece per operand (round 6: dF / decode of a handle created with mfma = "f16" | "bf16", unsplit on
    64 x 32 tiles): operands rounded once to fp16 / bf16, fp32 accumulate -- the error of a single rounding per operand (2^-11 / 2^-8 relative,
    times sqrt(K) in the sum), the same numbers on every run, and within that bound of the 64 x 64 one-piece kernel."""
    from ganmf_amd.engine import gemm_f32
    M, N, K = shape
    rng = np.random.RandomState(M + 3 * N + 7 * K)
    A, B, ref, bound_abs = _mk(rng, M, N, K, False, bkm)
    monkeypatch.setenv("GANMF_MFMA", mode)
    monkeypatch.setenv("GANMF_TUNE", "bf16w=2")
    monkeypatch.setenv("GANMF_DEBUG_PLAN", "1")
    capfd.readouterr()
    out, _ = gemm_f32(A, B, False, bkm)
    line = capfd.readouterr().err
    assert "64 x 32 tiles" in line and ("mfma " + mode) in line, line
    assert np.all(np.abs(out - ref) <= bound * bound_abs + 1e-30), float((np.abs(out - ref) / (bound_abs + 1e-30)).max())
    again, _ = gemm_f32(A, B, False, bkm)
    np.testing.assert_array_equal(out, again)
    monkeypatch.setenv("GANMF_TUNE", "bf16w=0")
    tiled, _ = gemm_f32(A, B, False, bkm)
    assert np.all(np.abs(tiled - out) <= 2 * bound * bound_abs + 1e-30)


@pytest.mark.parametrize("akm,bkm", [(False, False), (False, True)])
@pytest.mark.parametrize("shape", [(256, 50000, 33), (128, 17632, 32), (1, 2048, 1), (64, 2049, 10), (100, 5003, 64), (321, 4100, 7), (70, 3000, 63)])
def test_gemm_skinny_k_stream(akm, bkm, shape, monkeypatch):
    """gemm_skinny.hpp: K <= 64 behind a wide output (decode / dF of a narrow autoencoder, the generator product of a small
    num_factors) as a stream with fp32 FMAs on the way -- the default for such shapes; against the tiled kernels
    (GANMF_TUNE=skinny=0) and the fp64 product.  Ragged rows, columns and K; strips of 256, 128 and 64 columns."""
    from ganmf_amd.engine import gemm_f32
    M, N, K = shape
    rng = np.random.RandomState(M + N + K)
    A, B, ref, bound = _mk(rng, M, N, K, akm, bkm)
    out, _ = gemm_f32(A, B, akm, bkm)
    err = np.abs(out - ref)
    assert np.all(err <= 4e-7 * bound * np.sqrt(K) + 1e-30), float((err / (bound + 1e-30)).max())
    monkeypatch.setenv("GANMF_TUNE", "skinny=0")
    tiled, _ = gemm_f32(A, B, akm, bkm)
    assert np.all(np.abs(tiled - ref) <= 4e-7 * bound * np.sqrt(K) + 1e-30)
    assert np.all(np.abs(tiled - out) <= 8e-7 * bound * np.sqrt(K) + 1e-30)


@pytest.mark.parametrize("bkm", [False, True])
@pytest.mark.parametrize("shape", [(256, 32, 50001), (128, 32, 50000), (100, 17, 42001), (300, 32, 14100), (257, 31, 16385), (2049, 5, 2050)])
def test_gemm_skinny_n_stream(bkm, shape, monkeypatch, capfd):
    """gemm_skinny.hpp gemm_skinny_n_kernel: N <= 32 behind a long K (encode and dE at emb_dim 32: a [2B, 50 000] activation read once for
    a [2B, 32] result) as a stream on the fp32 MFMA, one 256-wide K slice per workgroup, slabs summed in split order -- the default for such
    shapes; against the fp64 product and the tiled kernels (GANMF_TUNE=skinny=0).  Ragged rows (more than one 256-row block too), fewer than
    32 columns, K that ends inside a slice and inside a 4-float group, both layouts of B; bitwise reproducible."""
    from ganmf_amd.engine import gemm_f32
    M, N, K = shape
    rng = np.random.RandomState(M + N + K)
    A, B, ref, bound = _mk(rng, M, N, K, False, bkm)
    monkeypatch.setenv("GANMF_DEBUG_PLAN", "1")
    capfd.readouterr()
    out, _ = gemm_f32(A, B, False, bkm)
    assert "skinny-N stream" in capfd.readouterr().err, "the planner did not select gemm_skinny_n_kernel for %r" % (shape,)
    err = np.abs(out - ref)
    assert np.all(err <= 4e-7 * bound * np.sqrt(K) + 1e-30), float((err / (bound + 1e-30)).max())
    again, _ = gemm_f32(A, B, False, bkm)
    assert np.array_equal(out, again)
    monkeypatch.setenv("GANMF_TUNE", "skinny=0")
    capfd.readouterr()
    tiled, _ = gemm_f32(A, B, False, bkm)
    assert "skinny-N stream" not in capfd.readouterr().err
    assert np.all(np.abs(tiled - out) <= 8e-7 * bound * np.sqrt(K) + 1e-30)
