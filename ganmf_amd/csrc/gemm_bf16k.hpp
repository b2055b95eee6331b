// The split-bf16 loop (gemm_bf16s.hpp: exact three-way split, six piece products, fp32 accumulate) in the 16-wave form of the
// fp32 ring kernel: ONE 64 x 64 tile per workgroup of 1024 threads = 4 K groups x (2 x 2 waves), K-tile 64 = four 16-wide
// chunks, chunk g of every K-tile to K group g.  For the step's K-heavy small-M products (256 / 128 rows, K = 1000 .. 3700,
// <= 256 workgroups: one per CU), where the 4-wave staged kernel is latency-bound (27 us against the fp32 ring kernel's 21 on the
// encode GEMM) and the fp32 ring kernel is MFMA-bound at the 2.0 GHz the chip holds (profiles/r03_gemm_stamps.md).
//   * staging: waves 0-7 own the A tile, waves 8-15 the B tile, one item of 8 elements per thread and K-tile (two 16-byte loads,
//     four pair splits, three 16-byte plane stores); the loads of PD = 2 K-tiles are in flight in registers, so the
//     fetch latency is covered without an LDS ring and without more workgroups per CU;
//   * two plane buffers (2 x 48 KiB) and ONE barrier per K-tile: in iteration t every wave reads its fragments of buffer t & 1 and
//     issues its six MFMAs, then splits K-tile t + 1 into buffer (t + 1) & 1 (last read in iteration t - 1, behind the barrier);
//   * per K-tile and SIMD: 4 x 6 MFMAs = 768 cycles of matrix pipe against 2 048 for the fp32 MFMA, beside ~800 cycles of VALU issue
//     for the splits; 144 KiB of LDS traffic (256 B/clk: 560 cycles);
//   * same pieces, same product order and the same two accumulators as gemm_bf16s_body; the K groups' partial sums meet in the
//     epilogue in group order (gemm_epilogue, KG = 4).
#pragma once
#include "gemm_bf16s.hpp"

namespace ganmf {

typedef float f32x4k __attribute__((ext_vector_type(4)));      // (a native vector: HIP's float4 is a struct, which inline asm takes through memory)

// One thread's item of a K-tile: 8 elements of ONE operand (waves 0-7: A, waves 8-15: B), two 16-byte loads.  The item's operand
// and layout (K-contiguous / K-major, the layouts of SplitStage<64, 64, KM>) are run-time, wave-uniform properties: address
// arithmetic and the pairing of the split branch on them, the loads and the plane stores themselves are the same instructions for
// every thread -- with the loads inside role branches hipcc cannot count them and waits with vmcnt(0) before every split, which
// serialises the whole prefetch (first build of this kernel: 34 us for the encode GEMM, 20 us without its fetches).
struct SplitItemK {
  const float* zp;
  const float* ptr;
  long long inc;      // floats per K-tile
  int aux, dst, ld;
  int dst1 = 0;      // K-major: dword offset of the item's second k-row RELATIVE to dst
  bool km;
  __device__ inline void init(bool km_, const float* __restrict__ base, int ld_, int r0, int rlimit, int kbeg, const float* zero, int w,
                              const int* __restrict__ gather = nullptr) {
    km = km_; ld = ld_;
    zp = zero + (w & 255) * 8;
    if (!km) {
      using S = SplitStage<64, 64, false>;
      const int row = w / S::CH, c8 = w % S::CH;
      const bool ok = (r0 + row) < rlimit;
      const int src_row = (ok && gather) ? gather[r0 + row] : r0 + row;      // (GemmP::a_gather: the embedding lookup rides in the fetch)
      ptr = ok ? base + (size_t)src_row * ld + kbeg + 8 * c8 : zp;
      inc = ok ? 64 : 0;
      aux = 0;
      dst = row * 32 + 4 * (c8 ^ S::swz(row));
      dst1 = 0;
    } else {      // (round 5) the [k][64 rows] bf16 image of SplitStage<64, 64, true>: four consecutive rows of two k-rows, two 8-byte stores per piece
      using S = SplitStage<64, 64, true>;
      const int kp = w / S::RQ, rq = w % S::RQ;
      const bool ok = (r0 + 4 * rq) < ld;
      ptr = base + (size_t)(kbeg + 2 * kp) * ld + r0 + 4 * rq;
      inc = (long long)64 * ld;
      aux = ok ? 2 * kp : -1;
      dst = (2 * kp) * S::KROW + 4 * ((rq >> 1) ^ S::swzk(2 * kp)) + 2 * (rq & 1);
      dst1 = (2 * kp + 1) * S::KROW + 4 * ((rq >> 1) ^ S::swzk(2 * kp + 1)) + 2 * (rq & 1) - dst;
    }
  }
  // the two source addresses of the item in the K-tile that starts kleft k's before the end of the range (no memory access here)
  __device__ inline void addresses(int kleft, const float*& s0, const float*& s1) {
    if (!km) { s0 = ptr; s1 = ptr + 4; }
    else {
      s0 = (aux >= 0 && aux < kleft) ? ptr : zp;
      s1 = (aux >= 0 && aux + 1 < kleft) ? ptr + ld : zp;
    }
    ptr += inc;
  }
  // `tok`: the token of the counted wait that covers this item's two loads (bf16k_wait_vm): the first instruction that reads a loaded
  // register carries it as an operand, so nothing of the split can be scheduled in front of the wait.  (Rounds 3-4 folded it into the
  // layout predicate of eight selects: a K-major item then paired its elements across the two k-rows it held.  With the [k][rows] image
  // an item of either layout is eight consecutive elements and the selects are gone: 8 of 71 vector instructions per wave and K-tile.)
  template <int NPIECE, bool F16>
  __device__ inline void split(const f32x4k (&v)[2], u32x4 (&pc)[3], const int tok, const float scale) const {
    // the four pairs of the item: 8 consecutive k of one row, or four rows of k-row k followed by the same four rows of k-row k + 1.  (Scalars, no private arrays: hipcc
    // parks those in LDS / scratch and then waits for every load where it is issued.)
    const f32x4k a = v[0], b = v[1];
    if constexpr (NPIECE == 3) {
      unsigned h, m, l;
      split_bf16x3_tok(a.x, a.y, h, m, l, tok); pc[0][0] = h; pc[1][0] = m; pc[2][0] = l;
      split_bf16x3_tok(a.z, a.w, h, m, l, tok); pc[0][1] = h; pc[1][1] = m; pc[2][1] = l;
      split_bf16x3_tok(b.x, b.y, h, m, l, tok); pc[0][2] = h; pc[1][2] = m; pc[2][2] = l;
      split_bf16x3_tok(b.z, b.w, h, m, l, tok); pc[0][3] = h; pc[1][3] = m; pc[2][3] = l;
    } else if constexpr (F16) {      // one IEEE fp16 piece after the exact power-of-two scale (static loss scaling per GEMM, gemm_f32.hpp)
      // (every element is multiplied by the scale before anything else reads it: the scale carries the token)
      const float sc = __uint_as_float(__float_as_uint(scale) | (unsigned)tok);
      pc[0][0] = cvt_pk_f16(a.x * sc, a.y * sc); pc[0][1] = cvt_pk_f16(a.z * sc, a.w * sc);
      pc[0][2] = cvt_pk_f16(b.x * sc, b.y * sc); pc[0][3] = cvt_pk_f16(b.z * sc, b.w * sc);
    } else {                         // one bf16 piece, round to nearest even
      pc[0][0] = cvt_pk_bf16_tok(a.x, a.y, tok); pc[0][1] = cvt_pk_bf16_tok(a.z, a.w, tok);
      pc[0][2] = cvt_pk_bf16_tok(b.x, b.y, tok); pc[0][3] = cvt_pk_bf16_tok(b.z, b.w, tok);
    }
  }
};

constexpr int BF16K_OPER = 3 * (64 * 64 / 2) * 2;      // dwords of one plane buffer: three pieces x (A + B) = 48 KiB
// ... of the loop with NPIECE pieces per operand: the one-piece forms (operands rounded to a single bf16 / fp16) file ONE plane per operand, 16 KiB
// per buffer -- their workgroup then needs the 64 KiB of the epilogue's four staged partial tiles, not 96 KiB, and two of them share a CU
// (round 6: the low-precision gUb + gV pair launch has more workgroups than CUs, as the fp32 pair_kernel<4, 2> does)
template <int NPIECE> constexpr int bf16k_oper() { return NPIECE * (64 * 64 / 2) * 2; }
template <int NPIECE> constexpr int bf16k_smem_dw() { return 2 * bf16k_oper<NPIECE>() > 4 * 64 * 64 ? 2 * bf16k_oper<NPIECE>() : 4 * 64 * 64; }
static_assert(bf16k_oper<3>() == BF16K_OPER && bf16k_smem_dw<3>() == 2 * BF16K_OPER && bf16k_smem_dw<1>() == 4 * 64 * 64, "plane buffers");
// K-tiles in flight in registers.  Three-piece loop (1.0 us per K-tile): TWO -- 2 us of fetch in flight cover the latency, and a launch that
// opens with half the requests per CU leaves more of the fabric to its neighbours' tails: the step + 1.4-1.8 % on the slower boxes against
// the four of rounds 3-4 (8 722 -> 8 875, 8 746 -> 8 868 in one-call alternations; three: + 1.0 %), + 2.7 % on a fast one (9 000-9 070 ->
// 9 230-9 310).  The one-piece loops (one MFMA per K-tile) likewise: DisGANMF at configs[4] 17 059 / 17 353 -> 17 283 / 17 576 steps/s in
// fp16, 17 281 / 17 524 -> 17 766 / 17 960 in bf16.
template <int NPIECE> constexpr int bf16k_pd() { return 2; }

// `s_waitcnt vmcnt(n)` (n wave-uniform, 0 .. 9) that hands out a token (always 0) every first use of the awaited registers is made to
// depend on (SplitItemK::split hands it to the first conversion of every pair as an operand), so that no consumer is scheduled above the wait.
// (A wait tied to the registers themselves, "+v", made hipcc COPY them in front of it on some paths: a read of a register whose load is
// still in flight.)
__device__ __forceinline__ int bf16k_wait_vm(int n, const f32x4k& v0, const f32x4k& v1) {
  int tok;
#define GANMF_BF16K_WAIT(N) asm volatile("s_waitcnt vmcnt(" #N ")\n\ts_mov_b32 %0, 0" : "=s"(tok) : "v"(v0), "v"(v1))
  if (n >= 8) { if (n >= 9) GANMF_BF16K_WAIT(9); else GANMF_BF16K_WAIT(8); }
  else if (n >= 4) { if (n >= 6) { if (n >= 7) GANMF_BF16K_WAIT(7); else GANMF_BF16K_WAIT(6); } else { if (n >= 5) GANMF_BF16K_WAIT(5); else GANMF_BF16K_WAIT(4); } }
  else { if (n >= 2) { if (n >= 3) GANMF_BF16K_WAIT(3); else GANMF_BF16K_WAIT(2); } else { if (n >= 1) GANMF_BF16K_WAIT(1); else GANMF_BF16K_WAIT(0); } }
#undef GANMF_BF16K_WAIT
  return tok;
}

// K loop of one 64 x 64 tile (tm, tn), K slice sp, batch bz -> acc (this wave's 32 x 32 block, K group kg's share of the sum).
// NPIECE = 3: the exact three-way split (fp32-accurate).  NPIECE = 1: every operand rounded to ONE bf16, or with F16 to one IEEE fp16
// after its power-of-two scale (MFMA_BF16 / MFMA_F16, gemm_f32.hpp): one MFMA per wave and K-tile, same staging and pipeline.
// at_entry(): called once, after the first PD K-tiles have been requested and have landed and before the first MFMA; it may issue
// EXTRA vector-memory loads of its own (the weight-gradient kernel's theta / m / v fetch), which the counted waits then allow for.
template <bool AKM, bool BKM, int NPIECE, bool F16, int EXTRA, class Entry>
__device__ __forceinline__ void bf16k_mainloop(const GemmP& p, const int tm, const int tn, const int sp, const int bz, float* __restrict__ smem,
                                               f32x16& acc_out, Entry&& at_entry) {
  static_assert(NPIECE == 3 || NPIECE == 1, "pieces per operand");
  static_assert(!F16 || NPIECE == 1, "fp16 pieces only in the single-piece mode");
  static_assert(2 * (bf16k_pd<NPIECE>() - 1) + EXTRA <= 9, "bf16k_wait_vm covers 0 .. 9");
  constexpr int BM = 64, BN = 64, BK = 64, PD = bf16k_pd<NPIECE>();
  const float sa = (F16 && p.a_scale != 0.f) ? p.a_scale : 1.f, sb = (F16 && p.b_scale != 0.f) ? p.b_scale : 1.f;
  using FA = SplitStage<64, 64, AKM>;
  using FB = SplitStage<64, 64, BKM>;
  unsigned* const buf = reinterpret_cast<unsigned*>(smem);      // [2][A planes x 3 | B planes x 3]
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kg = wave >> 2;
  const int wr = (wave >> 1) & 1, wc = wave & 1;
  const int li = lane & 31, lh = lane >> 5;
  const bool stage_a = wave < 8;      // (wave-uniform)
  const int m0 = tm * BM, n0 = tn * BN;
  const int kbeg = sp * p.k_per_split;
  const int kend = min(p.K, kbeg + p.k_per_split);
  const int nt = (kend - kbeg + BK - 1) / BK;

  f32x16 acc, accl;
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc[r] = 0.f; accl[r] = 0.f; }

  SplitItemK it;
  if (stage_a) it.init(AKM, p.A + (size_t)bz * p.a_batch_stride, p.lda, m0, p.M, kbeg, p.zero_page, tid,
                       (AKM || !p.a_gather) ? nullptr : p.a_gather + (size_t)bz * p.a_gather_batch);
  else it.init(BKM, p.B, p.ldb, n0, p.N, kbeg, p.zero_page, tid - 512);
  constexpr int OPER = bf16k_oper<NPIECE>();      // dwords of one plane buffer: [A planes x NPIECE | B planes x NPIECE]
  unsigned* const my_planes = buf + (stage_a ? 0 : NPIECE * FA::PLANE) + it.dst;      // this thread's 16 bytes of piece 0, buffer 0
  const float my_scale = stage_a ? sa : sb;

  f32x4k rg[PD][2];
#pragma unroll
  for (int s = 0; s < PD; ++s) { rg[s][0] = f32x4k{0.f, 0.f, 0.f, 0.f}; rg[s][1] = f32x4k{0.f, 0.f, 0.f, 0.f}; }      // (slots >= nt are never requested)
  int kleft = kend - kbeg;      // k's from the next tile to load to the end of the range
  auto load_tile = [&](auto ss) {
    constexpr int s = decltype(ss)::value;
    const float *s0, *s1;
    it.addresses(kleft, s0, s1);
    // issued from inline asm: hipcc cannot count register loads across the loop's back edge and waits with vmcnt(0) in front of
    // two splits out of three -- i.e. for the loads it has just issued -- which puts the whole fetch latency back into the loop
    // (18.5 us on the encode GEMM); an asm load is invisible to its wait insertion, the waits below are counted by hand
    f32x4k l0, l1;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(l0) : "v"(s0) : "memory");
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(l1) : "v"(s1) : "memory");
    rg[s][0] = l0; rg[s][1] = l1;
    kleft -= BK;
  };
  // `nwait` = vector-memory loads issued after the two of the K-tile in slot s (loads return in order)
  auto store_tile = [&](auto ss, int b, int nwait) {
    constexpr int s = decltype(ss)::value;
    const f32x4k v[2] = {rg[s][0], rg[s][1]};
    const int tok = bf16k_wait_vm(nwait, v[0], v[1]);
    u32x4 pc[3];
#ifdef GANMF_PERSIST_DIAG_BUILD
    if ((p.diag & 32) && stage_a) {      // timing only: the A-staging waves store unsplit bits (what pre-split activation planes would save)
      pc[0] = __builtin_bit_cast(u32x4, v[0]); pc[1] = __builtin_bit_cast(u32x4, v[1]); pc[2] = pc[0];
      pc[0][0] += (unsigned)tok;
    } else
#endif
    it.template split<NPIECE, F16>(v, pc, tok, my_scale);
    unsigned* o = my_planes + b * OPER;
    if (it.km) {      // (wave-uniform) K-major image: the item's two k-rows are two 8-byte stores per piece
      typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
      for (int q = 0; q < NPIECE; ++q) {
        *reinterpret_cast<u32x2*>(o + q * FA::PLANE) = u32x2{pc[q][0], pc[q][1]};
        *reinterpret_cast<u32x2*>(o + q * FA::PLANE + it.dst1) = u32x2{pc[q][2], pc[q][3]};
      }
    } else {
#pragma unroll
      for (int q = 0; q < NPIECE; ++q) *reinterpret_cast<u32x4*>(o + q * FA::PLANE) = pc[q];
    }
  };
  // prologue: K-tiles 0 .. PD-1 requested, tile 0 split into buffer 0
  static_for<0, PD>([&](auto ss) { if (decltype(ss)::value < nt) load_tile(ss); });
  store_tile(std::integral_constant<int, 0>{}, 0, 2 * (min(nt, PD) - 1));
  // everything requested so far lands before the loop is entered (the requests went out back to back with tile 0's, which the split
  // above has waited for): should hipcc move a prefetch register at the loop header, it moves a register that is complete
  if constexpr (PD == 4) asm volatile("s_waitcnt vmcnt(0)" :: "v"(rg[0][0]), "v"(rg[0][1]), "v"(rg[1][0]), "v"(rg[1][1]), "v"(rg[2][0]), "v"(rg[2][1]), "v"(rg[3][0]), "v"(rg[3][1]));
  else asm volatile("s_waitcnt vmcnt(0)" :: "v"(rg[0][0]), "v"(rg[0][1]), "v"(rg[1][0]), "v"(rg[1][1]));
  static_assert(PD == 4 || PD == 2, "the wait above names every prefetch register");
  at_entry();
  __syncthreads();
#ifdef GANMF_PERSIST_DIAG_BUILD
  if (p.stamps && tid == 0) p.stamps[(size_t)blockIdx.x * 4 + 1] = __builtin_amdgcn_s_memrealtime();      // first K-tile split into LDS
#endif

  auto step = [&](auto ss, int t) {      // K-tile t, whose registers were slot s = t % PD (consumed in iteration t - 1)
    constexpr int s = decltype(ss)::value, s1 = (s + 1) % PD;
    const unsigned* planes = buf + (t & 1) * OPER;
    u32x4 pa[3], pb[3];
#pragma unroll
    for (int q = 0; q < NPIECE; ++q) pa[q] = FA::frag(planes + q * FA::PLANE, wr * 32, kg, li, lh);
#pragma unroll
    for (int q = 0; q < NPIECE; ++q) pb[q] = FB::frag(planes + NPIECE * FA::PLANE + q * FB::PLANE, wc * 32, kg, li, lh);
    // the registers of tile t are free: request tile t + PD into them (before the MFMAs: the longer the fetch has)
#ifdef GANMF_PERSIST_DIAG_BUILD
    if (p.diag & 16) { if (t + PD < nt) kleft -= BK; } else      // timing only: no operand fetches after the prologue
#endif
    if (t + PD < nt) load_tile(ss);
    constexpr int ta[6] = {1, 0, 2, 1, 0, 0}, tb[6] = {1, 2, 0, 0, 1, 0};   // (mid,mid) (hi,lo) (lo,hi) (mid,hi) (hi,mid) (hi,hi): gemm_bf16s_body
#ifdef GANMF_PERSIST_DIAG_BUILD
    if (!(p.diag & 4))             // timing only: no MFMAs
#endif
    {
      if constexpr (NPIECE == 3) {
#pragma unroll
        for (int t6 = 0; t6 < 6; ++t6) {
          if (t6 < 5)
            accl = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, pa[ta[t6]]), __builtin_bit_cast(bf16x8, pb[tb[t6]]), accl, 0, 0, 0);
          else
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, pa[ta[t6]]), __builtin_bit_cast(bf16x8, pb[tb[t6]]), acc, 0, 0, 0);
        }
      } else if constexpr (F16) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, pa[0]), __builtin_bit_cast(f16x8, pb[0]), acc, 0, 0, 0);
      } else {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, pa[0]), __builtin_bit_cast(bf16x8, pb[0]), acc, 0, 0, 0);
      }
    }
#ifdef GANMF_PERSIST_DIAG_BUILD
    if (!(p.diag & 2))             // timing only: no split / plane stores after the prologue
#endif
    // under the MFMAs: split K-tile t + 1 into the other buffer.  Issued after its two loads: those of tiles t + 2 .. min(t + PD, nt - 1),
    // and the entry hook's EXTRA loads if tile t + 1 was requested in the prologue
    if (t + 1 < nt) store_tile(std::integral_constant<int, s1>{}, (t + 1) & 1, 2 * (min(t + PD, nt - 1) - (t + 1)) + (t + 1 < PD ? EXTRA : 0));
    __syncthreads();
  };
  for (int t = 0; t < nt; t += PD)
    static_for<0, PD>([&](auto ss) { if (t + decltype(ss)::value < nt) step(ss, t + decltype(ss)::value); });

  if constexpr (NPIECE == 3) acc += accl;
  if constexpr (F16) {
    if (sa != 1.f || sb != 1.f) acc *= 1.f / (sa * sb);      // undo the operand scaling in fp32 (powers of two: exact)
  }
  acc_out = acc;
}

template <bool AKM, bool BKM, int NPIECE = 3, bool F16 = false>
__device__ __forceinline__ void gemm_bf16k_body(const GemmP& p, const int bid, const int nblk, float* __restrict__ smem) {
  int tm, tn, sp, bz;
  tile_coords(p, bid, nblk, tm, tn, sp, bz);
  f32x16 acc[1][1];
#ifdef GANMF_PERSIST_DIAG_BUILD      // (GANMF_GEMM_STAMPS: only stand-alone launches hand out a stamp buffer, one slot per block of the grid)
  if (p.stamps && threadIdx.x == 0) p.stamps[(size_t)blockIdx.x * 4 + 0] = __builtin_amdgcn_s_memrealtime();
#endif
  bf16k_mainloop<AKM, BKM, NPIECE, F16, 0>(p, tm, tn, sp, bz, smem, acc[0][0], [] {});
#ifdef GANMF_PERSIST_DIAG_BUILD
  if (p.stamps && threadIdx.x == 0) p.stamps[(size_t)blockIdx.x * 4 + 2] = __builtin_amdgcn_s_memrealtime();
#endif
  static_assert(4 * 64 * 64 <= bf16k_smem_dw<NPIECE>(), "the workgroup's LDS must hold the four staged partial tiles");
  gemm_epilogue<64, 64, 1, 1, 4, AKM && BKM>(p, acc, smem, TileCoord{tm, tn, sp, bz, tm * 64, tn * 64});
#ifdef GANMF_PERSIST_DIAG_BUILD
  if (p.stamps) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); if (threadIdx.x == 0) p.stamps[(size_t)blockIdx.x * 4 + 3] = __builtin_amdgcn_s_memrealtime(); }
#endif
}

// (one-piece forms: 64 KiB of LDS, so two workgroups fit a CU -- if the CU admits their 32 waves: floor(800 / (ceil(sgpr / 16) * 16 + 16)) waves per SIMD
// is 8 only at <= 80 scalar registers (gemm_multi.hpp pair_kernel; the staged pass's batched generator launch has 5 452 workgroups))
template <bool AKM, bool BKM, int NPIECE = 3, bool F16 = false>
__global__ __launch_bounds__(1024) void gemm_bf16k_mfma(const GemmP p) {
  __shared__ __attribute__((aligned(16))) float smem[bf16k_smem_dw<NPIECE>()];      // three pieces: 96 KiB, one workgroup per CU; one piece: 64 KiB
  gemm_bf16k_body<AKM, BKM, NPIECE, F16>(p, (int)blockIdx.x, (int)gridDim.x, smem);
}

// (Round 6: the one-piece forms with their scalar registers capped at 80 -- two 64-KiB workgroups per CU admitted, as for pair_kernel -- measured 0.5-1 % SLOWER
// on configs[4]: their grids are one round anyway except the per-pass batched generator launch, and the spills cost the loop.  Not kept.)
template <int NPIECE, bool F16>
inline hipError_t gemm_bf16k_launch(hipStream_t st, const GemmP& p0, bool akm, bool bkm) {
  GemmP p = p0;
#ifdef GANMF_PERSIST_DIAG_BUILD
  if (getenv("GANMF_BF16K_DIAG")) p.diag = atoi(getenv("GANMF_BF16K_DIAG"));      // 2 no splits, 4 no MFMAs, 16 no fetches (wrong results)
#endif
  const int grid = p.tiles_m * p.tiles_n * p.nsplit * p.nbatch;
  if (grid <= 0) return hipSuccess;
#ifdef GANMF_PERSIST_DIAG_BUILD
  const bool stamping = gemm_stamps_on() && gemm_stamps_begin(p, grid, st);
  struct Report { const GemmP& p; int grid; hipStream_t st; bool akm, bkm, on; ~Report() { if (on) gemm_stamps_report(p, grid, st, akm, bkm, NPIECE == 3 ? "gemm_bf16k_mfma (split-bf16, 16 waves)" : "gemm_bf16k_mfma (one piece)"); } } report{p, grid, st, akm, bkm, stamping};
#endif
  {
    if (!akm && !bkm) GANMF_LAUNCH((gemm_bf16k_mfma<false, false, NPIECE, F16>), dim3(grid), dim3(1024), 0, st, p);
    else if (!akm && bkm) GANMF_LAUNCH((gemm_bf16k_mfma<false, true, NPIECE, F16>), dim3(grid), dim3(1024), 0, st, p);
    else if (akm && bkm) GANMF_LAUNCH((gemm_bf16k_mfma<true, true, NPIECE, F16>), dim3(grid), dim3(1024), 0, st, p);
    else return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

}  // namespace ganmf
