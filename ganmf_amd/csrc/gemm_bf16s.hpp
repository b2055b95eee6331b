// fp32 GEMM on the bf16 matrix cores, operands split ONCE when they are staged (gfx950 / CDNA4).
//
// Same contract as gemm_f32_mfma (GemmP, operand layouts NT / NN / TN, zero-page convention, split-K slabs,
// shared epilogue) with a different K loop:
//   * every fp32 operand element is split exactly into three bf16 pieces x = hi + mid + lo (gemm_f32.hpp,
//     split_bf16x3: round-to-nearest at each level) by the thread that LOADS it, and the pieces are written to
//     three LDS planes per operand; a.b is accumulated in fp32 from the six piece products of weight >= 2^-18
//     on v_mfma_f32_32x32x16_bf16.
//     (Splitting at fragment-read time instead — fp32 tiles in LDS, each wave splitting what it reads — repeats the
//     VALU work in both waves that share a fragment; hipcc emits VALU work and MFMAs in blocks, whose issue times
//     add on a SIMD (tools/ubench/mfma_valu.hip), and that variant was 3 % slower at 4096^3 and 30 % slower in
//     bf16 mode.)
//   * HBM/L2 -> registers -> split -> LDS.  One LDS buffer, the next K-tile prefetched into registers while the
//     current one is multiplied; two workgroups per CU cover each other's barriers.
//   * LDS plane layouts (dword = two bf16 with consecutive k, low half first):
//       K-contiguous operand  [row][BK/2], the 16-byte chunk (8 k's) index XOR-swizzled by the row
//                             -> one conflict-free ds_read_b128 per piece and fragment;
//       K-major operand       (round 5) [k][R] bf16 -- the operand's own orientation, two rows per dword -- with the 16-byte chunk
//                             (8 rows) index XORed by f(k & 3), read with the TRANSPOSING ds_read_b64_tr_b16: two reads per
//                             piece and fragment (k + 0..3 and k + 4..7 of the lane's eight) at 256 B/clk instead of four
//                             ds_read_b32 at 128 B/clk (rounds 1-4: [k/2][R] dwords of (k, k + 1) pairs), conflict-free; an item
//                             is four consecutive rows of TWO k-rows (two 16-byte loads whose lanes tile whole 256-byte row
//                             segments: a first version with 8 rows of one k-row per item -- 16-byte loads at a 32-byte lane
//                             stride -- tripled the time the NN kernels wait for memory), paired ALONG the row like a
//                             K-contiguous item and filed as two 8-byte stores per piece.
//     A lane (i = lane & 31, h = lane >> 5) holds k = 16c + 8h + 0..7 of MFMA step c for both operands.
//   * NPIECE = 1 rounds each operand to a single bf16 (RNE), or with F16 = true to a single fp16 after an exact
//     power-of-two scale and a clamp to the fp16 range: the mixed-precision modes (MFMA_BF16 / MFMA_F16).
#pragma once
#include "gemm_f32.hpp"

namespace ganmf {

template <int R, int BK, bool KM>
struct SplitStage {
  static constexpr int NW = R * BK / 8;       // work items (8 elements each) per K-tile
  static constexpr int NI = NW / 256;         // per thread
  static constexpr int PLANE = R * BK / 2;    // dwords per piece plane
  static constexpr int CH = BK / 8;           // K-contiguous: 16-byte chunks per row
  static constexpr int RQ = R / 4;            // K-major: row quads per k-row
  static constexpr int KROW = R / 2;          // K-major: dwords per k-row of a plane
  static_assert(R == 64 || R == 128, "K-major chunk swizzle is laid out for 64- and 128-row operand tiles");
  // K-major: XOR on the chunk index of k-row k.  A transposing read covers k-rows k0 .. k0 + 3 (k0 a multiple of four) x 16 rows per
  // 16-lane group, two groups per 32-lane half: with 128-byte k-rows (R = 64) rows k0 + 2, k0 + 3 fall on the banks of k0, k0 + 1 and
  // move to the other half of their 128 bytes; with 256-byte k-rows (R = 128) all four share their banks and take four different
  // 64-byte groups (the layout of round 4's wgrad_stream.hpp, SQ_LDS_BANK_CONFLICT 0)
  __device__ static inline int swzk(int k) { return R == 64 ? ((k & 3) >> 1) << 2 : (k & 3) << 2; }
  static_assert(NW % 256 == 0 && NI >= 1, "tile too small for 256 threads");
  static_assert(BK == 32 || BK == 64, "BK must be 32 or 64");

  __device__ static inline int swz(int row) { return BK == 32 ? (row >> 2) & 3 : (row >> 1) & 7; }

  const float* zp;          // this lane's 32-byte line of the zero page
  const float* ptr[NI];     // source of the next tile (KM: first of the two k-rows)
  int aux[NI];              // !KM: pointer increment per tile (0: zero-page lane); KM: first k-row of the item or -1
  int dst[NI];              // dword offset inside a plane (K-major: of the item's first k-row)
  int dst1[NI];             // K-major: of its second k-row

  __device__ inline void init(const float* __restrict__ base, int ld, int r0, int rlimit, int kbeg,
                              const float* zero, int tid) {
    zp = zero + (tid & 255) * 8;
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int w = j * 256 + tid;
      if constexpr (!KM) {
        const int row = w / CH, c8 = w % CH;
        const bool ok = (r0 + row) < rlimit;
        ptr[j] = ok ? base + (size_t)(r0 + row) * ld + kbeg + 8 * c8 : zp;
        aux[j] = ok ? BK : 0;
        dst[j] = row * (BK / 2) + 4 * (c8 ^ swz(row));
      } else {
        const int kp = w / RQ, rq = w % RQ;
        const bool ok = (r0 + 4 * rq) < ld;
        ptr[j] = base + (size_t)(kbeg + 2 * kp) * ld + r0 + 4 * rq;
        aux[j] = ok ? 2 * kp : -1;
        dst[j] = (2 * kp) * KROW + 4 * ((rq >> 1) ^ swzk(2 * kp)) + 2 * (rq & 1);
        dst1[j] = (2 * kp + 1) * KROW + 4 * ((rq >> 1) ^ swzk(2 * kp + 1)) + 2 * (rq & 1);
      }
    }
  }

  // fetch this thread's share of the tile that starts kleft k's before the end of the K range
  __device__ inline void load(float4 (&v)[NI][2], int ld, int kleft) {
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      if constexpr (!KM) {
        const float* s = ptr[j];
        v[j][0] = *reinterpret_cast<const float4*>(s);
        v[j][1] = *reinterpret_cast<const float4*>(s + 4);
        ptr[j] += aux[j];
      } else {
        const float* s0 = (aux[j] >= 0 && aux[j] < kleft) ? ptr[j] : zp;
        const float* s1 = (aux[j] >= 0 && aux[j] + 1 < kleft) ? ptr[j] + ld : zp;
        v[j][0] = *reinterpret_cast<const float4*>(s0);
        v[j][1] = *reinterpret_cast<const float4*>(s1);
        ptr[j] += (size_t)BK * ld;
      }
    }
  }

  // split / round and write the pieces to the planes at `planes` (piece q at planes + q * PLANE)
  bool diag_nosplit = false;
  template <int NPIECE, bool F16 = false>
  __device__ inline void store(unsigned* __restrict__ planes, const float4 (&v)[NI][2], float scale = 1.f) const {
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      // the four pairs of this item: 8 consecutive k of one row, or 8 consecutive rows of one k-row
      // (no private arrays here: hipcc parks them in scratch / LDS)
      const float4 a = v[j][0], b = v[j][1];
      u32x4 pc[3];
      auto one = [&](float x0, float x1, auto qq) {
        constexpr int q = decltype(qq)::value;
        if constexpr (NPIECE == 3) {
          unsigned h, m, l;
#ifdef GANMF_PERSIST_DIAG_BUILD
          if (diag_nosplit) { h = cvt_pk_bf16(x0, x1); m = h; l = h; } else      // timing only: one conversion instead of the 3-way split
#endif
          split_bf16x3(x0, x1, h, m, l);
          pc[0][q] = h; pc[1][q] = m; pc[2][q] = l;
        } else if constexpr (F16) {
          pc[0][q] = cvt_pk_f16(x0 * scale, x1 * scale);
        } else {
          pc[0][q] = cvt_pk_bf16(x0, x1);
        }
      };
      // (either layout pairs along its dwords: eight k of one row; or four rows of k-row k, then four rows of k-row k + 1)
      one(a.x, a.y, std::integral_constant<int, 0>{}); one(a.z, a.w, std::integral_constant<int, 1>{});
      one(b.x, b.y, std::integral_constant<int, 2>{}); one(b.z, b.w, std::integral_constant<int, 3>{});
#pragma unroll
      for (int q = 0; q < NPIECE; ++q) {
        if constexpr (KM) {
          typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
          *reinterpret_cast<u32x2*>(planes + q * PLANE + dst[j]) = u32x2{pc[q][0], pc[q][1]};
          *reinterpret_cast<u32x2*>(planes + q * PLANE + dst1[j]) = u32x2{pc[q][2], pc[q][3]};
        } else {
          *reinterpret_cast<u32x4*>(planes + q * PLANE + dst[j]) = pc[q];
        }
      }
    }
  }

  // one piece of the fragment of the 32-row block at tile row rb, MFMA step c, lane (i, h)
  __device__ static inline u32x4 frag(const unsigned* __restrict__ plane, int rb, int c, int i, int h) {
    if constexpr (!KM) {
      return *reinterpret_cast<const u32x4*>(plane + (rb + i) * (BK / 2) + 4 * ((2 * c + h) ^ swz(i)));
    } else {
      // ds_read_b64_tr_b16: lane 4 qq + pp of a 16-lane group supplies the address of k-row qq, rows 4 pp .. 4 pp + 3 of a 4 (k) x 16 (rows)
      // block and receives row (lane & 15) of the four k-rows.  Group gi = (i >> 4) takes rows 16 gi .. 16 gi + 15 of the 32-row block;
      // lane half h the k-rows 16 c + 8 h + 0..7, in two reads (+ 0..3, + 4..7: the same k & 3, hence the same chunk XOR).
      const int l16 = i & 15, gi = i >> 4, qq = l16 >> 2, pp = l16 & 3;
      const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) const unsigned*)plane;
      const unsigned addr = base + (unsigned)((16 * c + 8 * h + qq) * (KROW * 4) + ((((rb >> 3) + 2 * gi + (pp >> 1)) ^ swzk(qq)) << 4) + ((pp & 1) << 3));
#if defined(__HIP_DEVICE_COMPILE__)      // (LDS pointers are 32 bits wide in the device pass only)
      typedef __attribute__((address_space(3))) s16x4* lds_s16x4_p;
      const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)addr);
      const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(addr + 4 * (KROW * 4)));
      return __builtin_bit_cast(u32x4, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
#else
      (void)addr;
      return u32x4{0u, 0u, 0u, 0u};
#endif
    }
  }
};

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int BM, int BN, int BK, int NPIECE>
struct Bf16sLds {      // floats of LDS one workgroup needs: the piece planes, or the C tile the epilogue stages there
  static constexpr int OPER = NPIECE * (BM * BK / 2 + BN * BK / 2);
  static constexpr int DW = OPER > BM * BN ? OPER : BM * BN;
};

// (body = device function of (block index, blocks of this GEMM): one launch can carry more than one piece of work,
// gemm_multi.hpp; `smem` is the launch's only LDS object, Bf16sLds<...>::DW floats)
template <int BM, int BN, int BK, bool AKM, bool BKM, int NPIECE, bool F16 = false>
__device__ __forceinline__ void gemm_bf16s_body(const GemmP& p, const int bid, const int nblk, float* __restrict__ smem) {
  static_assert(!F16 || NPIECE == 1, "fp16 pieces only in the single-piece (mixed precision) mode");
  const float sa = (F16 && p.a_scale != 0.f) ? p.a_scale : 1.f, sb = (F16 && p.b_scale != 0.f) ? p.b_scale : 1.f;
  using SA = SplitStage<BM, BK, AKM>;
  using SB = SplitStage<BN, BK, BKM>;
  constexpr int WM = BM / 2, WN = BN / 2;
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr int NC = BK / 16;
  static_assert(NPIECE == 1 || NPIECE == 3, "pieces per operand");
  static_assert(BM % 32 == 0 && BN % 32 == 0, "row XOR of the K-major layout needs 32-row blocks");
  static_assert(Bf16sLds<BM, BN, BK, NPIECE>::OPER == NPIECE * (SA::PLANE + SB::PLANE), "LDS budget");
  unsigned* const planes_a = reinterpret_cast<unsigned*>(smem);
  unsigned* const planes_b = planes_a + NPIECE * SA::PLANE;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int li = lane & 31, lh = lane >> 5;

  int tm, tn, sp, bz;
  tile_coords(p, bid, nblk, tm, tn, sp, bz);
  const int m0 = tm * BM, n0 = tn * BN;
  const int kbeg = sp * p.k_per_split;
  const int kend = min(p.K, kbeg + p.k_per_split);
  const int nt = (kend - kbeg + BK - 1) / BK;
#ifdef GANMF_PERSIST_DIAG_BUILD
  // diagnostic build: 64 stamps per workgroup (thread 0): [0] entry, [1 + 5 it + j] K-tile `it` (it < 12): loads issued, MFMAs issued,
  // past the barrier, next planes stored, past the second barrier; [62] K loop done, [63] exit
#define GANMF_S_STAMP(i) do { if (p.stamps && tid == 0) p.stamps[(size_t)bid * 64 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define GANMF_S_STAMP(i) do { } while (0)
#endif
  GANMF_S_STAMP(0);

  f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  SA la;
  SB lb;
  la.init(p.A + (size_t)bz * p.a_batch_stride, p.lda, m0, p.M, kbeg, p.zero_page, tid);
  lb.init(p.B, p.ldb, n0, p.N, kbeg, p.zero_page, tid);
#ifdef GANMF_PERSIST_DIAG_BUILD
  la.diag_nosplit = lb.diag_nosplit = (p.diag & 2) != 0;
#endif

  float4 ra[SA::NI][2], rb[SB::NI][2];
  int kleft = kend - kbeg;
  la.load(ra, p.lda, kleft);
  lb.load(rb, p.ldb, kleft);
  kleft -= BK;
  la.template store<NPIECE, F16>(planes_a, ra, sa);
  lb.template store<NPIECE, F16>(planes_b, rb, sb);
  __syncthreads();

  u32x4 pa[2][TM][NPIECE], pb[2][TN][NPIECE];
  auto load_frags = [&](int set, int c) {
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int q = 0; q < NPIECE; ++q) pa[set][a][q] = SA::frag(planes_a + q * SA::PLANE, wr * WM + a * 32, c, li, lh);
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int q = 0; q < NPIECE; ++q) pb[set][b][q] = SB::frag(planes_b + q * SB::PLANE, wc * WN + b * 32, c, li, lh);
  };
  // Two accumulators per block in the split mode: the hi.hi products (weight 1) go to `acc`, the five correction
  // products (weights 2^-9 .. 2^-18) to `accl`, which stays ~2^-8 of `acc`, so the small terms are not rounded away
  // against a large running sum; the two are added once, before the epilogue.  With one accumulator the path missed
  // the two tightest Adam-amplified parity checks (5.6e-5 / 6.7e-5 against a 5e-5 bound the fp32 MFMA meets); with
  // two, every GEMM of the step can run on it and the whole GPU suite passes.  __launch_bounds__(256, 2) keeps the
  // 128 accumulator registers of the 128x128 tile within two workgroups per CU.
  f32x16 accl[NPIECE == 3 ? TM : 1][NPIECE == 3 ? TN : 1];
  if constexpr (NPIECE == 3) {
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int b = 0; b < TN; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) accl[a][b][r] = 0.f;
  }
  auto mfmas = [&](int set) {
    // piece products in increasing weight, blocks innermost (dependent MFMAs are TM*TN issues apart)
    constexpr int NT = NPIECE == 3 ? 6 : 1;
    constexpr int ta[6] = {1, 0, 2, 1, 0, 0}, tb[6] = {1, 2, 0, 0, 1, 0};   // (mid,mid) (hi,lo) (lo,hi) (mid,hi) (hi,mid) (hi,hi)
#pragma unroll
    for (int t6 = 0; t6 < NT; ++t6)
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) {
          const int ia = NT == 6 ? ta[t6] : 0, ib = NT == 6 ? tb[t6] : 0;
          if (NT == 6 && t6 < 5)
            accl[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, pa[set][a][ia]),
                                                                 __builtin_bit_cast(bf16x8, pb[set][b][ib]), accl[a][b], 0, 0, 0);
          else if constexpr (F16)
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, pa[set][a][ia]),
                                                               __builtin_bit_cast(f16x8, pb[set][b][ib]), acc[a][b], 0, 0, 0);
          else
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, pa[set][a][ia]),
                                                                __builtin_bit_cast(bf16x8, pb[set][b][ib]), acc[a][b], 0, 0, 0);
        }
  };

  for (int it = 0; it < nt; ++it) {
    const bool more = it + 1 < nt;
    if (more) {                      // next K-tile into registers while this one is multiplied
#ifdef GANMF_PERSIST_DIAG_BUILD
      if (!(p.diag & 16))            // timing only: the first K-tile's registers are reused (no operand traffic after the prologue)
#endif
      {
      la.load(ra, p.lda, kleft);
      lb.load(rb, p.ldb, kleft);
      }
      kleft -= BK;
    }
    if (it < 12) GANMF_S_STAMP(1 + 5 * it);
    load_frags(0, 0);
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      if (c + 1 < NC) load_frags((c + 1) & 1, c + 1);
#ifdef GANMF_PERSIST_DIAG_BUILD
      if (p.diag & 4) { if (c == 0) mfmas(0); continue; }      // timing only: one MFMA step per K-tile instead of NC
#endif
      mfmas(c & 1);
    }
    if (it < 12) GANMF_S_STAMP(2 + 5 * it);
    __syncthreads();                 // every wave has read its fragments: the planes may be overwritten
    if (it < 12) GANMF_S_STAMP(3 + 5 * it);
    if (more) {
      la.template store<NPIECE, F16>(planes_a, ra, sa);
      lb.template store<NPIECE, F16>(planes_b, rb, sb);
      if (it < 12) GANMF_S_STAMP(4 + 5 * it);
      __syncthreads();
      if (it < 12) GANMF_S_STAMP(5 + 5 * it);
    }
  }
  GANMF_S_STAMP(62);
  if constexpr (NPIECE == 3) {
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int b = 0; b < TN; ++b) acc[a][b] += accl[a][b];
  }
  if constexpr (F16) {
    if (sa != 1.f || sb != 1.f) {      // undo the operand scaling in fp32 (powers of two: exact)
      const float un = 1.f / (sa * sb);
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) acc[a][b] *= un;
    }
  }
  gemm_epilogue<BM, BN, TM, TN, 1, AKM && BKM>(p, acc, smem, TileCoord{tm, tn, sp, bz, m0, n0});
#ifdef GANMF_PERSIST_DIAG_BUILD
  if (p.stamps) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); }
#endif
  GANMF_S_STAMP(63);
}

template <int BM, int BN, int BK, bool AKM, bool BKM, int NPIECE, bool F16 = false>
__global__ __launch_bounds__(256, 2) void gemm_bf16s_mfma(const GemmP p) {
  __shared__ __attribute__((aligned(16))) float smem[Bf16sLds<BM, BN, BK, NPIECE>::DW];
  gemm_bf16s_body<BM, BN, BK, AKM, BKM, NPIECE, F16>(p, (int)blockIdx.x, (int)gridDim.x, smem);
}

template <int BM, int BN, int BK, int NPIECE, bool F16 = false>
inline hipError_t gemm_bf16s_launch(hipStream_t st, const GemmP& p, bool akm, bool bkm) {
  const int grid = p.tiles_m * p.tiles_n * p.nsplit * p.nbatch;
  if (grid <= 0) return hipSuccess;
  if (!akm && !bkm) GANMF_LAUNCH((gemm_bf16s_mfma<BM, BN, BK, false, false, NPIECE, F16>), dim3(grid), dim3(256), 0, st, p);
  else if (!akm && bkm) GANMF_LAUNCH((gemm_bf16s_mfma<BM, BN, BK, false, true, NPIECE, F16>), dim3(grid), dim3(256), 0, st, p);
  else if (akm && bkm) GANMF_LAUNCH((gemm_bf16s_mfma<BM, BN, BK, true, true, NPIECE, F16>), dim3(grid), dim3(256), 0, st, p);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}

template <int NPIECE, bool F16>
inline hipError_t gemm_bf16k_launch(hipStream_t st, const GemmP& p, bool akm, bool bkm);      // gemm_bf16k.hpp: the 16-wave form

inline hipError_t gemm_planes_launch(hipStream_t st, const GemmP& p, bool bkm);      // gemm_planes.hpp: the same plan on pre-split operands

inline hipError_t gemm_dispatch_staged(hipStream_t st, const GemmP& p, bool akm, bool bkm, const GemmPlan& pl) {
  if (pl.tile == 64 && pl.kg == 4) {
#ifdef GANMF_PERSIST_DIAG_BUILD      // experiment (profiles/r04_wgrad_stream.md): measured level with the in-loop split, not used by the step
    if (pl.mode == MFMA_BF16X3 && p.a_planes && p.b_planes && !akm && !p.a_gather) return gemm_planes_launch(st, p, bkm);
#endif
    if (pl.mode == MFMA_BF16X3) return gemm_bf16k_launch<3, false>(st, p, akm, bkm);
    if (pl.mode == MFMA_F16) return gemm_bf16k_launch<1, true>(st, p, akm, bkm);
    if (pl.mode == MFMA_BF16) return gemm_bf16k_launch<1, false>(st, p, akm, bkm);
  }
  if (pl.mode == MFMA_F16)
    return pl.tile == 128 ? gemm_bf16s_launch<128, 128, 32, 1, true>(st, p, akm, bkm) : gemm_bf16s_launch<64, 64, 64, 1, true>(st, p, akm, bkm);
  if (pl.mode == MFMA_BF16X3 && pl.tile == 64 && pl.bk == 32) return gemm_bf16s_launch<64, 64, 32, 3>(st, p, akm, bkm);
  if (pl.mode == MFMA_BF16X3)
    return pl.tile == 128 ? gemm_bf16s_launch<128, 128, 32, 3>(st, p, akm, bkm) : gemm_bf16s_launch<64, 64, 64, 3>(st, p, akm, bkm);
  return pl.tile == 128 ? gemm_bf16s_launch<128, 128, 32, 1>(st, p, akm, bkm) : gemm_bf16s_launch<64, 64, 64, 1>(st, p, akm, bkm);
}

}  // namespace ganmf
