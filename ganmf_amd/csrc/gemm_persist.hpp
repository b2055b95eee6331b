// Persistent fp32 MFMA GEMM for many-tile outputs with a short K (the scoring product U.V^T, GANMF.py:285-292).
//
// Same arithmetic, operand layouts, LDS images and zero-page convention as gemm_f32_mfma (gemm_f32.hpp); what changes
// is the life of a workgroup.  There, a workgroup computes ONE output tile: prologue burst, K walk, then the whole CU
// waits while the 64 KiB tile is staged through the ring and written out, and the two co-resident workgroups do all of
// that in phase (profiles/README.md, "In-kernel phase timing": prologue 4.2 us + store issue 12.1 us idle per 24.3 us
// K loop on the 6040 x 3706 x 250 product).  Here ONE workgroup per CU walks a list of output tiles:
//   * the K-tile ring (NS slots) runs ACROSS output tiles: the first K-tiles of the next output tile are in flight
//     while the current one finishes, so only the very first tile of a workgroup pays a prologue;
//   * the finished accumulators are dumped into a C staging area of their own (BM x BN floats next to the ring) and
//     the tile is written to HBM as whole rows, 16 B per lane, a few stores per K-step UNDER the next tile's MFMAs;
//   * vmcnt bookkeeping: stores and LDS-DMA share the counter in issue order.  The stores of a K-step are issued FIRST
//     in the step and the counted wait still counts only the LDS-DMA pieces of the younger K-tiles, i.e. it is never
//     larger than the number of operations that are guaranteed to have been issued behind the K-tile it waits for
//     (a masked-off store may not be issued at all): conservative by at most the stores of one step.
//   * tile order: the 8 XCDs (blockIdx % 8, observed dispatch; speed only) each own a rectangular block of the tile
//     grid, walked M-innermost, so an XCD's L2 holds its A band and streams its B panels once: operand fetch drops from
//     (#N panels) x |A| to about (xb_n x |A| + xb_m x |B|).
// LDS: NS * 32 KiB ring + BM*BN*4 staging = 160 KiB for 128 x 128 x 32, NS = 3: exactly one workgroup per CU.
#pragma once
#include <cstdio>
#include <cstdlib>

#include "gemm_f32.hpp"

namespace ganmf {

struct PersistP {
  int xb_m, xb_n;        // XCD blocking of the tile grid: xb_m * xb_n == 8
  int wgs_per_xcd;       // gridDim.x / 8
  int diag;              // timing-only experiments (wrong results): 1 no C stores, 2 no staging dump, 4 no K-tile refills after
                         // the prologue, 256 in-kernel stamps.  Only in a library built with -DGANMF_PERSIST_DIAG_BUILD
                         // (make DIAG=1): the shipped library ignores the GANMF_PERSIST_DIAG environment variable.
};

#ifdef GANMF_PERSIST_DIAG_BUILD
#define PERSIST_DIAG_BIT(q, bit) (((q).diag & (bit)) != 0)
inline int persist_diag_bits() {
  const char* dg = getenv("GANMF_PERSIST_DIAG");
  return dg ? atoi(dg) : 0;
}
#else
#define PERSIST_DIAG_BIT(q, bit) (false)
inline int persist_diag_bits() { return 0; }
#endif

// A float4 from LDS through a __restrict__ parameter.  hipcc's waitcnt pass puts `s_waitcnt vmcnt(0)` in front of every
// LDS read that MAY alias an LDS-DMA in flight; a read whose pointer carries alias-scope metadata (what inlining a
// function with a __restrict__ parameter leaves behind) is checked against the DMAs that carry scopes themselves -- the
// glds builtin carries none -- and gets no wait, exactly as the fragment reads of Stage::frag.  Without this the staging
// reads below drained the whole K-tile ring once per K-step (18 compiler-inserted vmcnt(0) in the first version's ISA).
__device__ inline float4 lds_read_f4(const float* __restrict__ p) { return *reinterpret_cast<const float4*>(p); }

// tiles [b0, b1) of `n` split into `parts` nearly equal contiguous ranges

template <int BM, int BN, int BK, int NS, int WGM, int WGN, bool AKM, bool BKM>
__global__ __launch_bounds__(64 * WGM * WGN) void gemm_f32_persist(const GemmP p, const PersistP q) {
  constexpr int NTHR = 64 * WGM * WGN;            // WGM x WGN waves, each owning (BM/WGM) x (BN/WGN) of the tile
  using SA = Stage<BM, BK, AKM, NTHR>;
  using SB = Stage<BN, BK, BKM, NTHR>;
  constexpr int WM = BM / WGM, WN = BN / WGN;
  constexpr int TM = WM / 32, TN = WN / 32;
  static_assert(TM >= 1 && TN >= 1 && WM % 32 == 0 && WN % 32 == 0, "wave tile must be whole 32x32 MFMA blocks");
  constexpr int BUF = SA::SZ + SB::SZ;
  constexpr int LOADS = SA::NP + SB::NP;
  constexpr int NC = BK / 8;
  constexpr int PPC = (LOADS + NC - 1) / NC;
  constexpr int C4 = BN / 4, RPP = NTHR / C4, NPIECE = BM / RPP;   // C rows are written as float4 pieces, NPIECE per thread
  static_assert(NS >= 3, "the ring must hold a K-tile of the next output tile while the current one finishes");
  __shared__ __attribute__((aligned(16))) float smem[NS * BUF + BM * BN];
  float* const cst = smem + NS * BUF;     // C staging, natural [BM][BN] image

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave / WGN, wc = wave % WGN;
  const int li = lane & 31, lh = lane >> 5;

  // ---- this workgroup's tile list: block (bx_m, bx_n) of the tile grid, local tiles j, j + W, j + 2W, ...
  const int x = blockIdx.x & 7, j = blockIdx.x >> 3, W = q.wgs_per_xcd;
  const int bx_m = x % q.xb_m, bx_n = x / q.xb_m;
  const int mb0 = part_begin(p.tiles_m, q.xb_m, bx_m), mb1 = part_begin(p.tiles_m, q.xb_m, bx_m + 1);
  const int nb0 = part_begin(p.tiles_n, q.xb_n, bx_n), nb1 = part_begin(p.tiles_n, q.xb_n, bx_n + 1);
  const int bm = mb1 - mb0, bt = bm * (nb1 - nb0);
  if (j >= bt) return;
  const int n_my = (bt - j + W - 1) / W;
  const int nt = (p.K + BK - 1) / BK;            // K-tiles per output tile
  const int total = n_my * nt;                   // K-steps of this workgroup
  auto tile_origin = [&](int l, int& m0, int& n0) {
    m0 = (mb0 + l % bm) * BM;
    n0 = (nb0 + l / bm) * BN;
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  // ---- issue side of the ring: which output tile / K offset the next K-tile to load belongs to
  SA la;
  SB lb;
  int issue_l = j;                 // local tile index the issue stream is in
  int kleft = p.K;                 // k's left in that tile from the next K-tile to issue (<= 0: nothing more to issue)
  bool reinit = false;             // the K-tile being refilled is the last of its output tile: move on after it
  auto issue_init = [&]() {
    int m0, n0;
    tile_origin(issue_l, m0, n0);
    la.init(p.A, p.lda, m0, p.M, 0, p.zero_page, tid);
    lb.init(p.B, p.ldb, n0, p.N, 0, p.zero_page, tid);
  };
  auto issue_advance = [&]() {     // after a K-tile has been handed to the refill: where does the next one come from
    kleft -= BK;
    if (kleft <= 0 && issue_l + W < bt) reinit = true;
  };
  issue_init();

  float4 fa[2][TM], fb[2][TN];
  auto load_frags = [&](int set, const float* __restrict__ tile, int c) {
#pragma unroll
    for (int a = 0; a < TM; ++a) fa[set][a] = SA::frag(tile, wr * WM + a * 32, c, li, lh);
#pragma unroll
    for (int b = 0; b < TN; ++b) fb[set][b] = SB::frag(tile + SA::SZ, wc * WN + b * 32, c, li, lh);
  };
  auto mfmas = [&](int set) {
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int b = 0; b < TN; ++b) {
        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[set][a].x, fb[set][b].x, acc[a][b], 0, 0, 0);
        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[set][a].y, fb[set][b].y, acc[a][b], 0, 0, 0);
        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[set][a].z, fb[set][b].z, acc[a][b], 0, 0, 0);
        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[set][a].w, fb[set][b].w, acc[a][b], 0, 0, 0);
      }
  };

  // ---- deferred C stores: rows of the staged tile, one float4 per lane; piece i of a thread = row tr + i*RPP.
  // A piece is READ from the staging area in an even chunk, next to that chunk's fragment reads, and WRITTEN in the
  // following odd chunk behind the lgkmcnt wait the MFMAs of that chunk need anyway: no LDS round trip of its own in the
  // MFMA stream (a first version read, waited and stored in one place: 3 exposed LDS latencies per K-step, 17 us of
  // 136).  The store is a raw buffer store: lanes outside the matrix get an out-of-range offset and are dropped by
  // the hardware, so the instruction is issued unconditionally -- no divergent branches, and the tile's columns
  // N .. ldc-1 (products with zero-page rows: exact zeros) are written like any other.
  const int tc = tid % C4, tr = tid / C4;
  int st_m0 = 0, st_n0 = 0, st_next = NPIECE;      // next piece to read; NPIECE: nothing pending
  const int per_step = std::min(2, (NPIECE + nt - 1) / nt);   // pieces per K-step (two staging registers); leftovers are flushed at the tile end
  const __amdgpu_buffer_rsrc_t c_rsrc =
      __builtin_amdgcn_make_buffer_rsrc((void*)p.C, (short)0, (int)((long long)p.M * p.ldc * 4), 0x00020000);
  float4 stv[2];
  int sto[2];
  bool sth[2] = {false, false};
  auto piece_read = [&](int k) {          // stage register k <- next piece (nothing if none is pending)
    if PERSIST_DIAG_BIT(q, 1) { st_next = NPIECE; return; }     // timing only
    if (st_next < NPIECE) {
      const int row_l = tr + st_next * RPP;
      const int row = st_m0 + row_l, col = st_n0 + tc * 4;
      stv[k] = lds_read_f4(cst + row_l * BN + tc * 4);
      sto[k] = (row < p.M && col < p.ldc) ? (row * p.ldc + col) * 4 : (int)0x80000000;
      sth[k] = true;
      ++st_next;
    }
  };
  auto piece_write = [&](int k) {
    if (sth[k]) {
      u32x4 bits;
      bits[0] = __float_as_uint(stv[k].x); bits[1] = __float_as_uint(stv[k].y);
      bits[2] = __float_as_uint(stv[k].z); bits[3] = __float_as_uint(stv[k].w);
      __builtin_amdgcn_raw_buffer_store_b128(bits, c_rsrc, sto[k], 0, 0);
      sth[k] = false;
    }
  };
  auto flush_pieces = [&]() {             // whatever is left of the staged tile, without overlap
    piece_write(0);
    piece_write(1);
    while (st_next < NPIECE) { piece_read(0); piece_write(0); }
  };

  // ---- prologue: the first NS K-tiles of the flattened K-step sequence
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    if (reinit) { issue_l += W; issue_init(); kleft = p.K; reinit = false; }
    if (kleft > 0) {
      la.template issue_range<0, SA::NP>(smem + s * BUF, p.lda, kleft, wave);
      lb.template issue_range<0, SB::NP>(smem + s * BUF + SA::SZ, p.ldb, kleft, wave);
      issue_advance();
    }
  }
  if (total >= NS) GANMF_WAIT_VMCNT((NS - 1) * LOADS);
  else GANMF_WAIT_VMCNT(0);
  __builtin_amdgcn_s_barrier();
  load_frags(0, smem, 0);

  int slot = 0, pend_slot = 0, pend_kleft = 0;
  int cur_l = j, kstep = 0;        // compute side: local tile index and K-step inside it
  auto refill_chunk = [&](auto cc) {
    constexpr int c = decltype(cc)::value;
    if (pend_kleft > 0 && !PERSIST_DIAG_BIT(q, 4)) {
      float* base = smem + pend_slot * BUF;
      constexpr int q0 = c * PPC < LOADS ? c * PPC : LOADS, q1 = (c + 1) * PPC < LOADS ? (c + 1) * PPC : LOADS;
      constexpr int a0 = q0 < SA::NP ? q0 : SA::NP, a1 = q1 < SA::NP ? q1 : SA::NP;
      constexpr int b0 = q0 > SA::NP ? q0 - SA::NP : 0, b1 = q1 > SA::NP ? q1 - SA::NP : 0;
      la.template issue_range<a0, a1>(base, p.lda, pend_kleft, wave);
      lb.template issue_range<b0, b1>(base + SA::SZ, p.ldb, pend_kleft, wave);
    }
  };
  int stamp_i = 0;
  auto step = [&](auto tail_tag) {
    constexpr bool TAIL = decltype(tail_tag)::value;
    if (PERSIST_DIAG_BIT(q, 256) && p.counters && (blockIdx.x == 0 || blockIdx.x == 101) && tid == 0 && stamp_i < 60)
      reinterpret_cast<unsigned long long*>(p.counters)[(blockIdx.x ? 64 : 0) + stamp_i++] = __builtin_amdgcn_s_memtime();
    const float* __restrict__ cur = smem + slot * BUF;
    const int nslot = (slot + 1 == NS) ? 0 : slot + 1;
    static_for<0, NC>([&](auto cc) {
      constexpr int c = decltype(cc)::value;
      if constexpr (c + 1 < NC) {
        load_frags((c + 1) & 1, cur, c + 1);
        if constexpr ((c & 1) == 0) { if (c / 2 < per_step) piece_read((c / 2) & 1); }
        refill_chunk(cc);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr ((c & 1) == 1) piece_write((c / 2) & 1);
      } else {
        refill_chunk(cc);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if constexpr (TAIL) GANMF_WAIT_VMCNT(0);
        else GANMF_WAIT_VMCNT((NS - 2) * LOADS);
        __builtin_amdgcn_s_barrier();
        // the slot just consumed is refilled during the next step with the next K-tile of the issue stream
        if (reinit) { issue_l += W; issue_init(); kleft = p.K; reinit = false; }
        pend_slot = slot;
        pend_kleft = kleft;
        if (kleft > 0) issue_advance();
        load_frags(0, smem + nslot * BUF, 0);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr ((c & 1) == 1) piece_write((c / 2) & 1);
      }
      mfmas(c & 1);
    });
    slot = nslot;
    if (++kstep == nt) {
      // ---- output tile complete: accumulators -> staging; its stores ride under the next tile's K-steps.
      // The staging area is normally free: two pieces per K-step are read in chunks 0 and 2 and written in chunks 1 and 3,
      // so a tile of NPIECE pieces drains in NPIECE / 2 K-steps and the end-of-step barrier of the last one separates the
      // reads from this dump.  A K range too short for that (nt < NPIECE / 2) flushes the rest here, behind a barrier.
      if (st_next < NPIECE || sth[0] || sth[1]) {      // uniform over the workgroup
        flush_pieces();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
      }
      if (!PERSIST_DIAG_BIT(q, 2))
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            cst[(wr * WM + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * BN + wc * WN + b * 32 + li] = acc[a][b][r];
            acc[a][b][r] = 0.f;
          }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      tile_origin(cur_l, st_m0, st_n0);
      st_next = 0;
      cur_l += W;
      kstep = 0;
    }
  };
  int s = 0;
  for (; s + NS - 1 < total; ++s) step(std::false_type{});
  for (; s < total; ++s) step(std::true_type{});
  flush_pieces();           // the last tile
}


// ---------------------------------------------------------------------------------------------------------------------
// Two persistent workgroups per CU (4 waves each, 128 x 128 tiles, 2-slot K ring): the form the measurements asked for.
// One 8-wave workgroup per CU puts both waves of every SIMD behind the SAME barrier: the bubble around it (last MFMAs
// draining, barrier, first fragment reads) is exposed in every K-step -- in-kernel stamps: 4750 cycles per 4096-cycle
// K-step with loads and stores switched off.  Two independent 4-wave workgroups cover each other's bubbles (4300 per
// step measured for the one-tile kernel's K loop), but 160 KiB / 2 leaves no room for a 64 KiB C staging area.  So:
//   * at the end of a tile the accumulators are COPIED to a second register set (64 VGPRs) and zeroed; the next tile's
//     K-steps start at once (its first two K-tiles are already in the ring);
//   * each following K-step dumps one eighth of the saved tile (each wave a 16 x 32 patch: 8 registers per lane) into one
//     half of a 16 KiB staging area, and the step after it -- behind the K loop's own barrier -- reads that half as
//     whole 128-byte rows and stores it (two float4 per thread, raw buffer stores, out-of-range lanes dropped by the
//     hardware).  Eight K-steps drain a tile: no barrier, no wait and no phase of its own for the C stores.
// LDS: 2 x 32 KiB ring + 2 x 8 KiB staging = 80 KiB.  VGPRs: two accumulator sets.
template <int BK, bool AKM, bool BKM>
__global__ __launch_bounds__(256, 2) void gemm_f32_persist2(const GemmP p, const PersistP q) {
  constexpr int BM = 128, BN = 128, NS = 2;
  using SA = Stage<BM, BK, AKM>;
  using SB = Stage<BN, BK, BKM>;
  constexpr int WM = 64, WN = 64, TM = 2, TN = 2;
  constexpr int BUF = SA::SZ + SB::SZ;
  constexpr int LOADS = SA::NP + SB::NP;
  constexpr int NC = BK / 8;
  constexpr int PPC = (LOADS + NC - 1) / NC;
  constexpr int UNITS = 8, HALF = 2048;            // dump units per tile; floats per staging half (4 patches of 16 x 32)
  static_assert(NC == 4, "the store pipeline below is laid out over four chunks per K-step");
  __shared__ __attribute__((aligned(16))) float smem[NS * BUF + 2 * HALF];
  float* const cst = smem + NS * BUF;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int li = lane & 31, lh = lane >> 5;
  const bool stamping = PERSIST_DIAG_BIT(q, 256) && p.counters && (blockIdx.x == 0 || blockIdx.x == 101) && tid == 0;
  unsigned long long* stamps = reinterpret_cast<unsigned long long*>(p.counters) + (blockIdx.x ? 64 : 0);
  if (stamping) stamps[60] = __builtin_amdgcn_s_memtime();

  const int x = blockIdx.x & 7, j = blockIdx.x >> 3, W = q.wgs_per_xcd;
  const int bx_m = x % q.xb_m, bx_n = x / q.xb_m;
  const int mb0 = part_begin(p.tiles_m, q.xb_m, bx_m), mb1 = part_begin(p.tiles_m, q.xb_m, bx_m + 1);
  const int nb0 = part_begin(p.tiles_n, q.xb_n, bx_n), nb1 = part_begin(p.tiles_n, q.xb_n, bx_n + 1);
  const int bm = mb1 - mb0, bt = bm * (nb1 - nb0);
  if (j >= bt) return;
  const int n_my = (bt - j + W - 1) / W;
  const int nt = (p.K + BK - 1) / BK;
  const int total = n_my * nt;
  auto tile_origin = [&](int l, int& m0, int& n0) {
    m0 = (mb0 + l % bm) * BM;
    n0 = (nb0 + l / bm) * BN;
  };

  f32x16 acc[TM][TN], accs[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc[a][b][r] = 0.f; accs[a][b][r] = 0.f; }

  SA la;
  SB lb;
  int issue_l = j, kleft = p.K;
  bool reinit = false;
  auto issue_init = [&]() {
    int m0, n0;
    tile_origin(issue_l, m0, n0);
    la.init(p.A, p.lda, m0, p.M, 0, p.zero_page, tid);
    lb.init(p.B, p.ldb, n0, p.N, 0, p.zero_page, tid);
  };
  auto issue_advance = [&]() {
    kleft -= BK;
    if (kleft <= 0 && issue_l + W < bt) reinit = true;
  };
  issue_init();

  float4 fa[2][TM], fb[2][TN];
  auto load_frags = [&](int set, const float* __restrict__ tile, int c) {
#pragma unroll
    for (int a = 0; a < TM; ++a) fa[set][a] = SA::frag(tile, wr * WM + a * 32, c, li, lh);
#pragma unroll
    for (int b = 0; b < TN; ++b) fb[set][b] = SB::frag(tile + SA::SZ, wc * WN + b * 32, c, li, lh);
  };
  auto mfmas = [&](int set) {
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int b = 0; b < TN; ++b) {
        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[set][a].x, fb[set][b].x, acc[a][b], 0, 0, 0);
        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[set][a].y, fb[set][b].y, acc[a][b], 0, 0, 0);
        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[set][a].z, fb[set][b].z, acc[a][b], 0, 0, 0);
        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[set][a].w, fb[set][b].w, acc[a][b], 0, 0, 0);
      }
  };

  // ---- the saved tile: dump unit u = (a, b, h): registers 8h .. 8h+7 of block (a, b) = rows 16h .. 16h+15 of that block
  int ds_unit = UNITS, ds_m0 = 0, ds_n0 = 0;       // next unit of `accs` to dump (UNITS: nothing saved / all dumped)
  int g = 0;                                       // K-steps done: the dump of step g goes to staging half g & 1
  bool rd_valid = false;                           // the other half holds a unit dumped in the previous step
  int rd_unit = 0, rd_m0 = 0, rd_n0 = 0;
  auto dump_regs = [&](auto uu, float* __restrict__ half) {
    constexpr int u = decltype(uu)::value, a = u >> 2, b = (u >> 1) & 1, h = u & 1;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int r = 8 * h + i;
      half[wave * 512 + ((r & 3) + 8 * ((r >> 2) & 1) + 4 * lh) * 32 + li] = accs[a][b][r];
    }
  };
  auto dump_unit = [&]() {         // (a switch on the unit: register arrays must be indexed by constants)
    float* half = cst + (g & 1) * HALF;
    switch (ds_unit) {
      case 0: dump_regs(std::integral_constant<int, 0>{}, half); break;
      case 1: dump_regs(std::integral_constant<int, 1>{}, half); break;
      case 2: dump_regs(std::integral_constant<int, 2>{}, half); break;
      case 3: dump_regs(std::integral_constant<int, 3>{}, half); break;
      case 4: dump_regs(std::integral_constant<int, 4>{}, half); break;
      case 5: dump_regs(std::integral_constant<int, 5>{}, half); break;
      case 6: dump_regs(std::integral_constant<int, 6>{}, half); break;
      case 7: dump_regs(std::integral_constant<int, 7>{}, half); break;
      default: break;
    }
  };
  const __amdgpu_buffer_rsrc_t c_rsrc =
      __builtin_amdgcn_make_buffer_rsrc((void*)p.C, (short)0, (int)((long long)p.M * p.ldc * 4), 0x00020000);
  float4 stv[2];
  int sto[2];
  bool sth[2] = {false, false};
  auto piece_read = [&](int k) {   // float4 number tid + 256 k of the staged unit: patch (wave that dumped it), row, 16-byte column
    if (rd_valid && !PERSIST_DIAG_BIT(q, 1)) {
      const int idx = tid + 256 * k;
      const int patch = idx >> 7, prow = (idx >> 3) & 15, c4 = idx & 7;
      const int a = rd_unit >> 2, b = (rd_unit >> 1) & 1, h = rd_unit & 1;
      const int row = rd_m0 + (patch >> 1) * 64 + a * 32 + h * 16 + prow;
      const int col = rd_n0 + (patch & 1) * 64 + b * 32 + c4 * 4;
      stv[k] = lds_read_f4(cst + ((g + 1) & 1) * HALF + patch * 512 + prow * 32 + c4 * 4);
      sto[k] = (row < p.M && col < p.ldc) ? (row * p.ldc + col) * 4 : (int)0x80000000;
      sth[k] = true;
    }
  };
  auto piece_write = [&](int k) {
    if (sth[k]) {
      u32x4 bits;
      bits[0] = __float_as_uint(stv[k].x); bits[1] = __float_as_uint(stv[k].y);
      bits[2] = __float_as_uint(stv[k].z); bits[3] = __float_as_uint(stv[k].w);
      __builtin_amdgcn_raw_buffer_store_b128(bits, c_rsrc, sto[k], 0, 0);
      sth[k] = false;
    }
  };
  auto after_dump = [&](bool dumped) {   // end of a K-step (behind its barrier): what the next step reads
    rd_valid = dumped;
    if (dumped) { rd_unit = ds_unit; rd_m0 = ds_m0; rd_n0 = ds_n0; ++ds_unit; }
    ++g;
  };
  auto flush_saved = [&]() {             // drain the saved tile outside the K loop (end of the walk; K ranges under 8 K-tiles)
    for (;;) {
      if (rd_valid) { piece_read(0); piece_read(1); piece_write(0); piece_write(1); rd_valid = false; }
      if (ds_unit >= UNITS) break;
      dump_unit();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      after_dump(true);
    }
  };

  // ---- prologue
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    if (reinit) { issue_l += W; issue_init(); kleft = p.K; reinit = false; }
    if (kleft > 0) {
      la.issue(smem + s * BUF, p.lda, kleft, wave);
      lb.issue(smem + s * BUF + SA::SZ, p.ldb, kleft, wave);
      issue_advance();
    }
  }
  if (total >= NS) GANMF_WAIT_VMCNT((NS - 1) * LOADS);
  else GANMF_WAIT_VMCNT(0);
  __builtin_amdgcn_s_barrier();
  load_frags(0, smem, 0);

  int slot = 0, pend_slot = 0, pend_kleft = 0;
  int cur_l = j, kstep = 0;
  auto refill_chunk = [&](auto cc) {
    constexpr int c = decltype(cc)::value;
    if (pend_kleft > 0) {
      float* base = smem + pend_slot * BUF;
      constexpr int q0 = c * PPC < LOADS ? c * PPC : LOADS, q1 = (c + 1) * PPC < LOADS ? (c + 1) * PPC : LOADS;
      constexpr int a0 = q0 < SA::NP ? q0 : SA::NP, a1 = q1 < SA::NP ? q1 : SA::NP;
      constexpr int b0 = q0 > SA::NP ? q0 - SA::NP : 0, b1 = q1 > SA::NP ? q1 - SA::NP : 0;
      la.template issue_range<a0, a1>(base, p.lda, pend_kleft, wave);
      lb.template issue_range<b0, b1>(base + SA::SZ, p.ldb, pend_kleft, wave);
    }
  };
  int stamp_i = 0;
  for (int s = 0; s < total; ++s) {
    if (PERSIST_DIAG_BIT(q, 256) && p.counters && (blockIdx.x == 0 || blockIdx.x == 101) && tid == 0 && stamp_i < 60)
      reinterpret_cast<unsigned long long*>(p.counters)[(blockIdx.x ? 64 : 0) + stamp_i++] = __builtin_amdgcn_s_memtime();
    const float* __restrict__ cur = smem + slot * BUF;
    const int nslot = slot ^ 1;
    const bool dumping = ds_unit < UNITS && !PERSIST_DIAG_BIT(q, 2);
    // chunk 0: read the first float4 of the unit dumped in the previous step
    load_frags(1, cur, 1);
    piece_read(0);
    refill_chunk(std::integral_constant<int, 0>{});
    __builtin_amdgcn_sched_barrier(0);
    mfmas(0);
    // chunk 1: store it, read the second
    load_frags(0, cur, 2);
    piece_read(1);
    refill_chunk(std::integral_constant<int, 1>{});
    __builtin_amdgcn_sched_barrier(0);
    piece_write(0);
    mfmas(1);
    // chunk 2: store the second; dump this step's unit of the saved tile into the other half
    load_frags(1, cur, 3);
    if (dumping) dump_unit();
    refill_chunk(std::integral_constant<int, 2>{});
    __builtin_amdgcn_sched_barrier(0);
    piece_write(1);
    mfmas(0);
    // chunk 3: K-tile s+1 has landed for every wave, slot `slot` is free, this step's dump is visible
    refill_chunk(std::integral_constant<int, 3>{});
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    GANMF_WAIT_VMCNT(0);
    __builtin_amdgcn_s_barrier();
    if (reinit) { issue_l += W; issue_init(); kleft = p.K; reinit = false; }
    pend_slot = slot;
    pend_kleft = kleft;
    if (kleft > 0) issue_advance();
    load_frags(0, smem + nslot * BUF, 0);
    __builtin_amdgcn_sched_barrier(0);
    mfmas(1);
    after_dump(dumping);
    slot = nslot;
    if (++kstep == nt) {
      // output tile complete: save it in the second register set (draining what an even shorter K range left behind)
      if (ds_unit < UNITS) flush_saved();
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
          for (int r = 0; r < 16; ++r) { accs[a][b][r] = acc[a][b][r]; acc[a][b][r] = 0.f; }
      tile_origin(cur_l, ds_m0, ds_n0);
      ds_unit = 0;
      cur_l += W;
      kstep = 0;
    }
  }
  if (stamping) stamps[61] = __builtin_amdgcn_s_memtime();
  // ---- the last tile of the walk: nothing is left to hide its stores under, but the K ring is idle now -- dump every
  // remaining unit at once into it (unit u at u * HALF: 8 x 8 KiB = the 64 KiB ring), one barrier, then 2 float4 per unit
  if (rd_valid) { piece_read(0); piece_read(1); piece_write(0); piece_write(1); rd_valid = false; }
  if (ds_unit < UNITS) {
    const int u0 = ds_unit;
    float* const big = smem;               // every K-tile has been consumed (loop-end barrier) and no LDS-DMA is in flight
    static_for<0, UNITS>([&](auto uu) {
      constexpr int u = decltype(uu)::value;
      if (u >= u0) dump_regs(uu, big + u * HALF);
    });
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int u = u0; u < UNITS; ++u) {
      const int a = u >> 2, b = (u >> 1) & 1, h = u & 1;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int idx = tid + 256 * k;
        const int patch = idx >> 7, prow = (idx >> 3) & 15, c4 = idx & 7;
        const int row = ds_m0 + (patch >> 1) * 64 + a * 32 + h * 16 + prow;
        const int col = ds_n0 + (patch & 1) * 64 + b * 32 + c4 * 4;
        const float4 v = lds_read_f4(big + u * HALF + patch * 512 + prow * 32 + c4 * 4);
        u32x4 bits;
        bits[0] = __float_as_uint(v.x); bits[1] = __float_as_uint(v.y); bits[2] = __float_as_uint(v.z); bits[3] = __float_as_uint(v.w);
        __builtin_amdgcn_raw_buffer_store_b128(bits, c_rsrc, (row < p.M && col < p.ldc) ? (row * p.ldc + col) * 4 : (int)0x80000000, 0, 0);
      }
    }
    ds_unit = UNITS;
  }
  if (stamping) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); stamps[62] = __builtin_amdgcn_s_memtime(); }
}

// host side -----------------------------------------------------------------------------------------------------------
struct PersistPlan {
  int xb_m = 8, xb_n = 1, grid = 256;
  int rounds = 0;          // output tiles of the busiest workgroup
  double fetch_mb = 0;     // operand bytes the 8 L2s fetch under this blocking
};

// choose the XCD blocking (xb_m x xb_n = 8): fewest rounds first, then least operand fetch
inline PersistPlan persist_plan(int M, int N, int K, int tile, int cus) {
  PersistPlan best;
  best.rounds = 1 << 30;
  const int tm = (M + tile - 1) / tile, tn = (N + tile - 1) / tile;
  const int W = std::max(1, cus / 8);
  for (int xm = 1; xm <= 8; xm *= 2) {
    const int xn = 8 / xm;
    int rounds = 0;
    double fetch = 0;
    for (int i = 0; i < xm; ++i)
      for (int jn = 0; jn < xn; ++jn) {
        const int bm = part_begin(tm, xm, i + 1) - part_begin(tm, xm, i), bn = part_begin(tn, xn, jn + 1) - part_begin(tn, xn, jn);
        rounds = std::max(rounds, (bm * bn + W - 1) / W);
        fetch += 4.0 * K * tile * (bm + bn);
      }
    if (rounds < best.rounds || (rounds == best.rounds && fetch < best.fetch_mb * 1e6)) {
      best.rounds = rounds; best.xb_m = xm; best.xb_n = xn; best.fetch_mb = fetch / 1e6;
    }
  }
  best.grid = 8 * W;
  return best;
}

template <int BM, int BN, int BK, int NS, int WGM, int WGN>
inline hipError_t gemm_persist_launch(hipStream_t st, const GemmP& p, bool akm, bool bkm, const PersistPlan& pp) {
  PersistP q{pp.xb_m, pp.xb_n, pp.grid / 8, persist_diag_bits()};
  if (akm || bkm) return hipErrorInvalidValue;     // only the NT product (scores) takes this route
  if PERSIST_DIAG_BIT(q, 256) {      // diagnostic build path: per-K-step s_memtime stamps of two workgroups, printed after the launch
    static unsigned long long* dbg = nullptr;
    if (!dbg) { if (hipMalloc((void**)&dbg, 128 * 8) != hipSuccess) return hipErrorOutOfMemory; }
    (void)hipMemset(dbg, 0, 128 * 8);
    GemmP pd = p;
    pd.counters = reinterpret_cast<unsigned*>(dbg);
    GANMF_LAUNCH((gemm_f32_persist<BM, BN, BK, NS, WGM, WGN, false, false>), dim3(pp.grid), dim3(64 * WGM * WGN), 0, st, pd, q);
    (void)hipDeviceSynchronize();
    unsigned long long hs[128];
    (void)hipMemcpy(hs, dbg, sizeof hs, hipMemcpyDeviceToHost);
    static int printed = 0;
    if (printed++ < 2)
      for (int b = 0; b < 2; ++b) {
        fprintf(stderr, "[persist stamps wg %d] cycles per K-step:", b ? 101 : 0);
        for (int i = 1; i < 60 && hs[b * 64 + i]; ++i) fprintf(stderr, " %llu", hs[b * 64 + i] - hs[b * 64 + i - 1]);
        fprintf(stderr, "\n");
      }
    return hipGetLastError();
  }
  GANMF_LAUNCH((gemm_f32_persist<BM, BN, BK, NS, WGM, WGN, false, false>), dim3(pp.grid), dim3(64 * WGM * WGN), 0, st, p, q);
  return hipGetLastError();
}

// Eligibility: plain-store NT product on the fp32 MFMA, 128 x 128 tiles, no split-K / batching, and enough output
// tiles that every CU gets at least three (below that the one-tile-per-workgroup kernel fills the chip better).
inline bool gemm_persist_eligible(const GemmP& p, bool akm, bool bkm, const GemmPlan& pl, int force) {
  if (akm || bkm || p.epi.kind != EPI_STORE || p.nbatch > 1 || pl.nsplit != 1 || pl.mode != MFMA_F32 || pl.tile != 128) return false;
  if (!p.c_pad_writable || (long long)p.M * p.ldc * 4 >= (1LL << 31)) return false;   // buffer stores: 32-bit offsets, pad columns zeroed
  if (force == 0) return false;
  if (force > 0) return true;     // 1: eight waves, 2: four waves (run_gemm copies the value into pl.persist)
  return (long long)pl.tiles_m * pl.tiles_n >= 3LL * GEMM_CUS;
}

inline hipError_t gemm_dispatch_persist(hipStream_t st, const GemmP& p, bool akm, bool bkm, const GemmPlan& pl) {
  const PersistPlan pp = persist_plan(p.M, p.N, p.K, 128, GEMM_CUS);
  // 8 waves (two per SIMD, 64 x 32 wave tiles): one wave's LDS-DMA issue and barrier waits hide under its partner's MFMAs
  // (one wave per SIMD measured 85 % of the MFMA rate on the K loop, two 95 %: profiles/README.md); pl.persist == 2
  // selects the 4-wave form for A/B runs
  if (pl.persist == 2) return gemm_persist_launch<128, 128, 32, 3, 2, 2>(st, p, akm, bkm, pp);
  if (pl.persist == 3) return gemm_persist_launch<128, 128, 32, 3, 2, 4>(st, p, akm, bkm, pp);
  // default: two persistent 4-wave workgroups per CU
  if (akm || bkm) return hipErrorInvalidValue;
  const PersistPlan p2 = persist_plan(p.M, p.N, p.K, 128, 2 * GEMM_CUS);
  PersistP q{p2.xb_m, p2.xb_n, p2.grid / 8, persist_diag_bits()};
  if PERSIST_DIAG_BIT(q, 256) {      // diagnostic: per-K-step s_memtime stamps of two workgroups, printed after the launch
    static unsigned long long* dbg = nullptr;
    if (!dbg) { if (hipMalloc((void**)&dbg, 128 * 8) != hipSuccess) return hipErrorOutOfMemory; }
    (void)hipMemset(dbg, 0, 128 * 8);
    GemmP pd = p;
    pd.counters = reinterpret_cast<unsigned*>(dbg);
    GANMF_LAUNCH((gemm_f32_persist2<32, false, false>), dim3(p2.grid), dim3(256), 0, st, pd, q);
    (void)hipDeviceSynchronize();
    unsigned long long hs[128];
    (void)hipMemcpy(hs, dbg, sizeof hs, hipMemcpyDeviceToHost);
    static int printed = 0;
    if (printed++ < 1)
      for (int b = 0; b < 2; ++b) {
        fprintf(stderr, "[persist2 stamps wg %d] cycles per K-step:", b ? 101 : 0);
        for (int i = 1; i < 60 && hs[b * 64 + i]; ++i) fprintf(stderr, " %llu", hs[b * 64 + i] - hs[b * 64 + i - 1]);
        fprintf(stderr, "\n   entry -> first K-step %llu, last K-step start -> flush %llu, flush + store drain %llu, whole workgroup %llu\n",
                hs[b * 64 + 0] - hs[b * 64 + 60], hs[b * 64 + 61] - hs[b * 64 + 23], hs[b * 64 + 62] - hs[b * 64 + 61], hs[b * 64 + 62] - hs[b * 64 + 60]);
      }
    return hipGetLastError();
  }
  GANMF_LAUNCH((gemm_f32_persist2<32, false, false>), dim3(p2.grid), dim3(256), 0, st, p, q);
  return hipGetLastError();
}

}  // namespace ganmf
