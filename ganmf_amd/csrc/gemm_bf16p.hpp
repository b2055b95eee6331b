// Persistent fp32-accurate NT GEMM on the bf16 matrix cores from PRE-SPLIT operands (gfx950 / CDNA4): the scoring product
// U[ids] . V^T of _compute_item_score (GANMF.py:285-292) at many-tile shapes (ML-1M: 6040 x 3706 x 250).
//
// Same arithmetic as gemm_bf16s.hpp's MFMA_BF16X3 mode -- every fp32 operand element is the exact sum of three bf16 pieces
// x = hi + mid + lo (round to nearest at each level, split_bf16x3), a.b is accumulated in fp32 from the six piece products of
// weight >= 2^-18 on v_mfma_f32_32x32x16_bf16, hi.hi in one accumulator and the five corrections in a second -- but the split
// is done ONCE per element by a pre-pass (presplit_rows_kernel: the embedding gather of the scored rows rides in it) instead of
// once per consuming workgroup inside the K loop, and the GEMM's life is the persistent one of gemm_persist.hpp:
//   * operands arrive as three bf16 planes per matrix, [piece][row][K/2] dwords (a dword = two consecutive k), rows padded with
//     zero rows to whole 128-row tiles and K to whole 32-k tiles, so the kernel needs no zero page and no bounds on its reads;
//   * one 8-wave workgroup per CU (two waves per SIMD, wave tile 64 x 32) walks a list of 128 x 128 output tiles; the K-tile
//     ring (2 slots of 48 KiB: 3 planes x 128 rows x 64 B for A and for B) runs ACROSS tiles and is filled by LDS-DMA
//     (global_load_lds_dwordx4, 1 KiB per wave and instruction, six per wave and K-tile, all issued at the start of the K-step
//     that precedes their use), in the image the fragment reads of SplitStage<.., false> expect: [row][16 dwords], the 16-byte
//     chunk index XOR-swizzled by the row on the SOURCE address -- one conflict-free ds_read_b128 per piece and fragment;
//   * the K loop is MFMA + fragment reads only: no VALU split, no register staging; per K-tile and wave 24 MFMAs (768 cycles)
//     behind 18 KiB of fragment reads;
//   * a finished tile goes to a 64 KiB staging area (hi and correction accumulators added first) and leaves as whole 512-byte
//     rows, one float4 per thread and K-step, under the next tile's K-steps (raw buffer stores: out-of-range lanes are dropped
//     by the hardware); the 8 XCDs each own a rectangle of the tile grid (persist_plan).
// LDS: 2 x 48 KiB + 64 KiB = 160 KiB: exactly one workgroup per CU.
#pragma once
#include "gemm_bf16s.hpp"
#include "gemm_persist.hpp"

namespace ganmf {

constexpr int BF16P_TILE = 128, BF16P_BK = 32;

// One split pass for up to two matrices (blocks [0, blocks0): job 0, the rest: job 1).  Rows [0, nrows) of `src` (row r = src +
// (ids ? ids[r] : r) * ld, K valid floats) -> three bf16 planes in TILE-MAJOR order, the exact LDS image the GEMM's K loop reads:
//   planes[((tile * nt + kt) * 3 + piece) * 2048 + row * 16 + 4 * (chunk ^ swz(row)) + d]
// for row block `tile` (128 rows), K-tile kt (32 k = 16 dwords of two consecutive k each), row inside the block, 16-byte chunk
// of the row's 64 bytes, dword d of the chunk: one K-tile of one operand is 24 KiB of CONTIGUOUS memory, so the GEMM's loads are
// whole cache lines (a [piece][row][K/2] layout hands every wave-level load sixteen half-used lines: measured 13 B/clk per CU of
// operand ingest, 3 660 cycles per K-step against 1 536 of MFMA).  Rows >= nrows and k >= K are zero.  One thread per dword.
struct PresplitJob {
  const float* src;
  int ld;
  const int* ids;
  int nrows, rows_pad;
  unsigned* planes;
};
__global__ __launch_bounds__(256) void presplit_rows_kernel(const PresplitJob j0, const PresplitJob j1, int blocks0, int K, int kp2) {
  const bool second = (int)blockIdx.x >= blocks0;
  const PresplitJob& j = second ? j1 : j0;
  const int bx = second ? (int)blockIdx.x - blocks0 : (int)blockIdx.x, nbx = second ? (int)gridDim.x - blocks0 : blocks0;
  const long long total = (long long)j.rows_pad * kp2;
  const int nt = kp2 / 16;
  for (long long i = (long long)bx * blockDim.x + threadIdx.x; i < total; i += (long long)nbx * blockDim.x) {
    const int r = (int)(i / kp2), k2 = (int)(i % kp2);      // (reads coalesce along k; the tiled writes land as 64-byte runs)
    float x0 = 0.f, x1 = 0.f;
    if (r < j.nrows) {
      const float* s = j.src + (size_t)(j.ids ? j.ids[r] : r) * j.ld;
      if (2 * k2 < K) x0 = s[2 * k2];
      if (2 * k2 + 1 < K) x1 = s[2 * k2 + 1];
    }
    unsigned h, m, l;
    split_bf16x3(x0, x1, h, m, l);
    const int tile = r >> 7, row = r & 127, kt = k2 >> 4, dw = k2 & 15;
    const int chunk = dw >> 2, d = dw & 3;
    const size_t at = ((size_t)(tile * nt + kt) * 3) * 2048 + row * 16 + 4 * (chunk ^ ((row >> 2) & 3)) + d;
    j.planes[at] = h; j.planes[at + 2048] = m; j.planes[at + 4096] = l;
  }
}

struct Bf16pP {
  const unsigned* A;      // tile-major planes of the scored rows (presplit_rows_kernel): [a_rows_pad / 128][nt][3][128][16]
  const unsigned* B;      // ... of the other factor
  int a_rows_pad, b_rows_pad, kp2;
  float* C;
  int ldc, M, N;          // rows >= M and columns >= ldc of a tile are not written
  int tiles_m, tiles_n, nt;      // nt = K-tiles per output tile (kp2 / 16)
  int xb_m, xb_n, wgs_per_xcd;
  int late_mask;                 // experiment: which waves take the MFMA-first order (0: waves NW/2 ..)
  unsigned long long* dbg;       // diagnostic builds (make DIAG=1, GANMF_BF16P_STAMPS=1): s_memtime sums per loop phase of workgroup 0
};

template <int WGM, int WGN>
__global__ __launch_bounds__(64 * WGM * WGN) void gemm_bf16p_persist(const Bf16pP p) {
  constexpr int BM = BF16P_TILE, BN = BF16P_TILE, BK = BF16P_BK, NS = 2;
  constexpr int NW = WGM * WGN, NTHR = 64 * NW;
  constexpr int WM = BM / WGM, WN = BN / WGN, TM = WM / 32, TN = WN / 32;
  static_assert(TM >= 1 && TN >= 1, "wave tile must be whole 32x32 MFMA blocks");
  using SF = SplitStage<BM, BK, false>;                   // fragment addressing of a [128][16 dwords] plane
  constexpr int PLANE = BM * BK / 2;                      // dwords per piece plane (8 KiB)
  constexpr int OPER = 3 * PLANE;                         // one operand's three planes
  constexpr int BUF = 2 * OPER;                           // ring slot: A planes then B planes (48 KiB)
  constexpr int PIECES = BUF * 4 / 1024;                  // 1 KiB LDS-DMA pieces per K-tile (48)
  static_assert(PIECES % NW == 0, "pieces per wave");
  constexpr int LOADS = PIECES / NW;                      // 16-byte chunks per thread and K-tile
  constexpr int C4 = BN / 4, RPP = NTHR / C4, NPIECE = BM / RPP;
  __shared__ __attribute__((aligned(16))) unsigned smem[NS * BUF + BM * BN];
  float* const cst = reinterpret_cast<float*>(smem + NS * BUF);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave / WGN, wc = wave % WGN;
  const int li = lane & 31, lh = lane >> 5;

  const int x = blockIdx.x & 7, j = blockIdx.x >> 3, W = p.wgs_per_xcd;
  const int bx_m = x % p.xb_m, bx_n = x / p.xb_m;
  const int mb0 = part_begin(p.tiles_m, p.xb_m, bx_m), mb1 = part_begin(p.tiles_m, p.xb_m, bx_m + 1);
  const int nb0 = part_begin(p.tiles_n, p.xb_n, bx_n), nb1 = part_begin(p.tiles_n, p.xb_n, bx_n + 1);
  const int bm = mb1 - mb0, bt = bm * (nb1 - nb0);
  if (j >= bt) return;
  const int n_my = (bt - j + W - 1) / W;
  const int nt = p.nt;
  const int total = n_my * nt;
  auto tile_origin = [&](int l, int& m0, int& n0) {
    m0 = (mb0 + l % bm) * BM;
    n0 = (nb0 + l / bm) * BN;
  };

  f32x16 acc[TM][TN], accl[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc[a][b][r] = 0.f; accl[a][b][r] = 0.f; }

  // ---- issue side.  A K-tile is 3 072 16-byte chunks (48 KiB: A's three planes, then B's); thread t stages chunks J * NTHR + t,
  // J = 0 .. LOADS - 1, through registers.  The operands are stored tile-major in exactly this image (presplit_rows_kernel), so
  // chunk c of A comes from 16 c bytes into the (row block, K-tile) record and goes to 16 c bytes into the ring slot: linear,
  // whole cache lines.  (LDS-DMA, the fp32 kernels' route, was measured at the same speed: the ingest limit was the layout.)
  const unsigned* src[LOADS];      // per chunk: source dword address at the next K-tile to load
  u32x4 stage[LOADS];              // the K-tile in flight
  int issue_l = j, ktiles_left = nt;
  bool reinit = false;
  auto issue_init = [&]() {
    int m0, n0;
    tile_origin(issue_l, m0, n0);
#pragma unroll
    for (int J = 0; J < LOADS; ++J) {
      const int c = J * NTHR + tid;
      const bool isb = c >= OPER / 4;
      const unsigned* base = isb ? p.B : p.A;
      const int tile = (isb ? n0 : m0) >> 7;
      src[J] = base + (size_t)tile * nt * OPER + (isb ? c - OPER / 4 : c) * 4;
    }
  };
  auto load_begin = [&]() { if (reinit) { issue_l += W; issue_init(); ktiles_left = nt; reinit = false; } };
  auto load_one = [&](auto JJ) {    // chunk J of the next K-tile of the issue stream -> registers
    constexpr int J = decltype(JJ)::value;
    stage[J] = *reinterpret_cast<const u32x4*>(src[J]);
    src[J] += OPER;
  };
  auto load_end = [&]() { if (--ktiles_left == 0 && issue_l + W < bt) reinit = true; };
  auto load_tile = [&]() {
    load_begin();
    static_for<0, LOADS>([&](auto JJ) { load_one(JJ); });
    load_end();
  };
  auto write_one = [&](auto JJ, unsigned* slot_base) {      // registers -> ring slot
    constexpr int J = decltype(JJ)::value;
    *reinterpret_cast<u32x4*>(slot_base + (J * NTHR + tid) * 4) = stage[J];
  };
  auto write_tile = [&](unsigned* slot_base) { static_for<0, LOADS>([&](auto JJ) { write_one(JJ, slot_base); }); };
  issue_init();

  u32x4 pa[2][TM][3], pb[2][TN][3];
  auto load_frags = [&](int set, const unsigned* __restrict__ tile, int c) {
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int q = 0; q < 3; ++q) pa[set][a][q] = SF::frag(tile + q * PLANE, wr * WM + a * 32, c, li, lh);
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int q = 0; q < 3; ++q) pb[set][b][q] = SF::frag(tile + OPER + q * PLANE, wc * WN + b * 32, c, li, lh);
  };
  // The MFMAs of one chunk (piece products in increasing weight, blocks innermost: (mid,mid) (hi,lo) (lo,hi) (mid,hi) (hi,mid) ->
  // accl, (hi,hi) -> acc); side(i) is issued behind MFMA i and pinned there.  In-kernel stamps (make DIAG=1, GANMF_BF16P_STAMPS):
  // per K-step 555 + 837 cycles of LDS issue phases (fragment reads; plane writes + fragment reads) and 840 of barrier skew beside
  // 855 of MFMA -- the two waves of a SIMD move in lockstep behind the workgroup's one barrier, so nothing covers the LDS phases
  // (208 KiB per K-step at the 128 B/clk the CU's LDS delivers).  Dealing the fragment reads, plane writes and operand loads out
  // over the MFMA gaps was tried (97 instead of 86 us: a stalled LDS or VMEM issue then holds back the wave's next MFMA while its
  // SIMD partner is stalled in the same place).
  constexpr int NMFMA = 6 * TM * TN;
  auto mfmas_with = [&](auto set_c, auto&& side) {
    constexpr int set = decltype(set_c)::value;
    constexpr int ta[6] = {1, 0, 2, 1, 0, 0}, tb[6] = {1, 2, 0, 0, 1, 0};
    static_for<0, NMFMA>([&](auto ii) {
      constexpr int i = decltype(ii)::value, t6 = i / (TM * TN), a = (i % (TM * TN)) / TN, b = i % TN;
      if constexpr (t6 < 5)
        accl[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, pa[set][a][ta[t6]]),
                                                             __builtin_bit_cast(bf16x8, pb[set][b][tb[t6]]), accl[a][b], 0, 0, 0);
      else
        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, pa[set][a][ta[t6]]),
                                                            __builtin_bit_cast(bf16x8, pb[set][b][tb[t6]]), acc[a][b], 0, 0, 0);
      side(ii);
      __builtin_amdgcn_sched_barrier(0);
    });
  };

  // ---- deferred C stores (gemm_persist.hpp): piece i of a thread = row tr + i * RPP of the staged tile, one float4 per lane;
  // read from the staging area in the first chunk of a K-step, written at the top of the next K-step
  const int tc = tid % C4, tr = tid / C4;
  int st_m0 = 0, st_n0 = 0, st_next = NPIECE;
  const __amdgpu_buffer_rsrc_t c_rsrc =
      __builtin_amdgcn_make_buffer_rsrc((void*)p.C, (short)0, (int)((long long)p.M * p.ldc * 4), 0x00020000);
  constexpr int NPS = (NPIECE + 7) / 8;        // pieces per K-step: a tile drains within eight K-steps (K = 250: one tile's K range)
  float4 stv[NPS];
  int sto[NPS];
  bool sth[NPS];
#pragma unroll
  for (int k = 0; k < NPS; ++k) { sto[k] = 0; sth[k] = false; }
  auto piece_read = [&]() {
#pragma unroll
    for (int k = 0; k < NPS; ++k)
      if (st_next < NPIECE) {
        const int row_l = tr + st_next * RPP;
        const int row = st_m0 + row_l, col = st_n0 + tc * 4;
        stv[k] = lds_read_f4(cst + row_l * BN + tc * 4);
        sto[k] = (row < p.M && col < p.ldc) ? (row * p.ldc + col) * 4 : (int)0x80000000;
        sth[k] = true;
        ++st_next;
      }
  };
  auto piece_write = [&]() {
#pragma unroll
    for (int k = 0; k < NPS; ++k)
      if (sth[k]) {
        u32x4 bits;
        bits[0] = __float_as_uint(stv[k].x); bits[1] = __float_as_uint(stv[k].y);
        bits[2] = __float_as_uint(stv[k].z); bits[3] = __float_as_uint(stv[k].w);
        __builtin_amdgcn_raw_buffer_store_b128(bits, c_rsrc, sto[k], 0, 0);
        sth[k] = false;
      }
  };
  auto pending = [&]() { bool any = false; for (int k = 0; k < NPS; ++k) any |= sth[k]; return any; };
  auto flush_pieces = [&]() {
    piece_write();
    while (st_next < NPIECE) { piece_read(); piece_write(); }
  };

  // ---- prologue: K-tiles 0 and 1 into the ring, K-tile 2 into the staging registers
  int loaded = 0;                              // K-tiles fetched so far
  load_tile(); ++loaded;
  write_tile(smem);
  if (total > 1) { load_tile(); ++loaded; write_tile(smem + BUF); }
  if (total > 2) { load_tile(); ++loaded; }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  load_frags(0, smem, 0);

  int slot = 0, cur_l = j, kstep = 0;
  const bool late = p.late_mask ? ((p.late_mask >> wave) & 1) != 0 : wave >= NW / 2;            // (uniform per wave: a scalar branch)
#ifdef GANMF_PERSIST_DIAG_BUILD
  unsigned long long ph[6] = {0, 0, 0, 0, 0, 0}, tprev = 0;
  const bool stamping = p.dbg != nullptr && blockIdx.x == 8 && (tid == 0 || tid == 64);
#define BF16P_STAMP(i) if (stamping) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); ph[i] += t_ - tprev; tprev = t_; }
  if (stamping) tprev = __builtin_amdgcn_s_memtime();
#else
#define BF16P_STAMP(i)
#endif
  for (int s = 0; s < total; ++s) {
    const unsigned* __restrict__ cur = smem + slot * BUF;
    const int nslot = slot ^ 1;
    // The two waves of a SIMD (waves w and w + NW / 2) take the two halves of each chunk in OPPOSITE order: the early half issues its
    // LDS / VMEM work and then its MFMAs, the late half its MFMAs first -- so one wave's fragment reads, plane writes and operand
    // loads run under its SIMD partner's MFMAs although both sit behind the same workgroup barrier (in lockstep, stamps showed
    // 555 + 837 cycles of LDS phases and 840 of barrier skew per K-step beside 855 of MFMA).
    // chunk 0 (k 0..15): the piece of the drained tile that was read one K-step ago is stored first and the next one leaves the
    // staging area; fragments of chunk 1
    piece_write();
    piece_read();
    if (!late) load_frags(1, cur, 1);
    __builtin_amdgcn_sched_barrier(0);
    BF16P_STAMP(0)
    mfmas_with(std::integral_constant<int, 0>{}, [](auto) {});
    if (late) { load_frags(1, cur, 1); __builtin_amdgcn_sched_barrier(0); }
    BF16P_STAMP(1)
    // chunk 1 (k 16..31): once its fragments are in registers nothing reads slot `slot` any more
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    BF16P_STAMP(2)
    __builtin_amdgcn_s_barrier();              // every wave is done with slot `slot`; K-tile s+1 (written one K-step ago) is visible
    BF16P_STAMP(3)
    auto refill = [&]() {
      if (s + 2 < total) {                     // K-tile s+2 has been in flight for a whole K-step: into the freed slot ...
        write_tile(smem + slot * BUF);
        if (loaded < total) { load_tile(); ++loaded; }      // ... and K-tile s+3 takes its place in the registers
      }
      if (s + 1 < total) load_frags(0, smem + nslot * BUF, 0);
    };
    if (!late) refill();
    __builtin_amdgcn_sched_barrier(0);
    BF16P_STAMP(4)
    mfmas_with(std::integral_constant<int, 1>{}, [](auto) {});
    if (late) { refill(); __builtin_amdgcn_sched_barrier(0); }
    BF16P_STAMP(5)
    slot = nslot;
    if (++kstep == nt) {
      // ---- output tile complete: hi + corrections -> staging; its stores ride under the next tile's K-steps
      if (st_next < NPIECE || pending()) {           // (a K range shorter than NPIECE K-tiles: drain what is left; uniform over the workgroup)
        flush_pieces();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
      }
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            cst[(wr * WM + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * BN + wc * WN + b * 32 + li] = acc[a][b][r] + accl[a][b][r];
            acc[a][b][r] = 0.f; accl[a][b][r] = 0.f;
          }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      tile_origin(cur_l, st_m0, st_n0);
      st_next = 0;
      cur_l += W;
      kstep = 0;
    }
  }
  flush_pieces();           // the last tile
#ifdef GANMF_PERSIST_DIAG_BUILD
  if (stamping) { for (int i = 0; i < 6; ++i) p.dbg[(tid ? 8 : 0) + i] = ph[i]; p.dbg[(tid ? 8 : 0) + 6] = total; }
#endif
}

inline bool bf16p_eligible(int M, int N, int K) {
  // many-tile NT products with a short K: from three 128 x 128 tiles per CU (below that the one-tile kernels fill the chip better)
  const long long tiles = (long long)((M + BF16P_TILE - 1) / BF16P_TILE) * ((N + BF16P_TILE - 1) / BF16P_TILE);
  return tiles >= 3LL * GEMM_CUS && K >= 1 && (long long)M * ((N + 63) / 64 * 64) * 4 < (1LL << 31);
}

// C[M, ldc] = A_planes . B_planes^T on stream st; the planes were written by presplit_rows_kernel
inline hipError_t gemm_bf16p_launch(hipStream_t st, const unsigned* a_planes, int a_rows_pad, const unsigned* b_planes,
                                    int b_rows_pad, int kp2, float* C, int ldc, int M, int N) {
  Bf16pP q{};
  q.A = a_planes; q.B = b_planes; q.a_rows_pad = a_rows_pad; q.b_rows_pad = b_rows_pad; q.kp2 = kp2;
  q.C = C; q.ldc = ldc; q.M = M; q.N = N;
  q.tiles_m = (M + BF16P_TILE - 1) / BF16P_TILE; q.tiles_n = (N + BF16P_TILE - 1) / BF16P_TILE;
  q.nt = kp2 / (BF16P_BK / 2);
  if (q.tiles_m * BF16P_TILE > a_rows_pad || q.tiles_n * BF16P_TILE > b_rows_pad || kp2 % (BF16P_BK / 2)) return hipErrorInvalidValue;
  const PersistPlan pp = persist_plan(M, N, 2 * kp2, BF16P_TILE, GEMM_CUS);
  q.xb_m = pp.xb_m; q.xb_n = pp.xb_n; q.wgs_per_xcd = pp.grid / 8;
#ifdef GANMF_PERSIST_DIAG_BUILD      // experiment switches (make DIAG=1): the 4-wave form and the per-wave phase order were measured and not kept
  static const int waves = [] { const char* e = getenv("GANMF_BF16P_WAVES"); return e ? atoi(e) : 8; }();
  static const int late_mask = [] { const char* e = getenv("GANMF_BF16P_LATE"); return e ? (int)strtol(e, nullptr, 0) : 0; }();
#else
  constexpr int waves = 8, late_mask = 0;
#endif
  q.late_mask = late_mask;
#ifdef GANMF_PERSIST_DIAG_BUILD
  static unsigned long long* dbg = nullptr;
  static int printed = 0;
  if (getenv("GANMF_BF16P_STAMPS")) {
    if (!dbg) { if (hipMalloc((void**)&dbg, 16 * 8) != hipSuccess) return hipErrorOutOfMemory; }
    (void)hipMemset(dbg, 0, 16 * 8);
    q.dbg = dbg;
  }
#endif
  if (waves == 4) GANMF_LAUNCH((gemm_bf16p_persist<2, 2>), dim3(pp.grid), dim3(256), 0, st, q);
  else GANMF_LAUNCH((gemm_bf16p_persist<2, 4>), dim3(pp.grid), dim3(512), 0, st, q);
#ifdef GANMF_PERSIST_DIAG_BUILD
  if (q.dbg && printed++ < 3) {
    (void)hipDeviceSynchronize();
    unsigned long long hs[16];
    (void)hipMemcpy(hs, dbg, sizeof hs, hipMemcpyDeviceToHost);
    for (int w = 0; w < 2; ++w)
      fprintf(stderr, "[bf16p stamps wave %d, %llu K-steps] per K-step: top->frags issued %llu | mfmas(0) %llu | lgkm wait %llu | barrier %llu | "
              "write+load+frags %llu | mfmas(1) %llu   (s_memtime ticks)\n", w, hs[8 * w + 6],
              hs[8 * w + 0] / hs[8 * w + 6], hs[8 * w + 1] / hs[8 * w + 6], hs[8 * w + 2] / hs[8 * w + 6], hs[8 * w + 3] / hs[8 * w + 6],
              hs[8 * w + 4] / hs[8 * w + 6], hs[8 * w + 5] / hs[8 * w + 6]);
  }
#endif
  return hipGetLastError();
}

}  // namespace ganmf
