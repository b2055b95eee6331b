// The 16-wave split-bf16 loop (gemm_bf16k.hpp) on a 64 x 32 tile with a 128-deep K-tile: the form for the generator step's M = 128
// products behind a wide output (dF = dE . We^T: 2 x 116 tiles), which the 64 x 64 kernel can only spread over the chip as two K
// slices per tile + a slab-sum launch (round-4 verdict, item 1).  Half the output per workgroup, so the product runs UNSPLIT on 232
// workgroups, and per k of the K range a workgroup spends less than the 64 x 64 kernel on everything that kernel's iteration adds up
// from (profiles/r05_bf16k_phases.md):
//   * 1024 threads = 8 K groups x (2 x 1 waves): K-tile 128 = eight 16-wide chunks, chunk g to K group g; a wave owns ONE 32 x 32 block
//     (16 accumulator registers + 16 for the small products), reads six 16-byte fragments and issues six MFMAs per K-tile -- per k half
//     of the fragment reads, barriers and MFMA issues of the 64 x 64 form;
//   * staging: (64 + 32) x 128 elements per K-tile = 12 per thread, the same for every thread: eight consecutive k of an A row (two
//     16-byte loads, four pair splits, one 16-byte store per piece; 16 threads along a row's 512 bytes) and four consecutive k of a B row
//     (one load, two pair splits, one 8-byte store per piece); per k three quarters of the 64 x 64 form's split work;
//   * PD = 2 K-tiles in flight in registers (24 registers), loads from inline asm behind hand-counted waits whose token the first
//     conversion of every pair carries (the technique and its lint: gemm_bf16k.hpp, tools/check_asm_prefetch.py).  (PD = 3 / 4: the
//     stand-alone product level / 0.5-1 us slower, the step 1.2 % / 2.2 % slower than with PD = 2 -- 2.8 us of fetch are in flight
//     either way, and a launch that opens with fewer requests per CU leaves more of the fabric to its neighbours' tails);
//   * two plane buffers of 3 x (64 + 32) x 128 bf16 = 72 KiB, one barrier per K-tile; rows of 256 bytes with the 16-byte chunk index
//     XOR-ed with the row (conflict-free ds_read_b128 / ds_write_b64);
//   * same pieces, same six products in the same order as gemm_bf16s_body / gemm_bf16k; the eight K groups' partial sums meet in
//     gemm_epilogue's row pass in group order.
// A K-contiguous; B K-contiguous (NT: dF) or K-major (NN: decode -- its image is [k][32 n] and the fragments come from the transposing
// ds_read_b64_tr_b16); no gather / CSR / planes, nsplit = 1: gemm_bf16w_eligible.
#pragma once
#include "gemm_bf16k.hpp"

namespace ganmf {

constexpr int BF16W_BM = 64, BF16W_BN = 32, BF16W_BK = 128, BF16W_KG = 8, BF16W_PD = 2;
constexpr int BF16W_PA = BF16W_BM * BF16W_BK / 2;              // dwords of one piece plane of A
constexpr int BF16W_PB = BF16W_BN * BF16W_BK / 2;
constexpr int BF16W_OPER = 3 * (BF16W_PA + BF16W_PB);          // dwords of one plane buffer: 72 KiB

// `s_waitcnt vmcnt(n)`, n in {0, 3, 6, 9} (whole K-tiles of three loads issued behind the awaited one), handing out the token of bf16k_wait_vm
__device__ __forceinline__ int bf16w_wait_vm(int n, const f32x4k& v0, const f32x4k& v1, const f32x4k& v2) {
  int tok;
#define GANMF_BF16W_WAIT(N) asm volatile("s_waitcnt vmcnt(" #N ")\n\ts_mov_b32 %0, 0" : "=s"(tok) : "v"(v0), "v"(v1), "v"(v2))
  if (n >= 9) GANMF_BF16W_WAIT(9);
  else if (n >= 6) GANMF_BF16W_WAIT(6);
  else if (n >= 3) GANMF_BF16W_WAIT(3);
  else GANMF_BF16W_WAIT(0);
#undef GANMF_BF16W_WAIT
  return tok;
}

// dwords of one plane buffer of the loop with NPIECE pieces per operand, and of the workgroup's LDS (the eight K groups' partial tiles meet there)
template <int NPIECE> constexpr int bf16w_oper() { return NPIECE * (BF16W_PA + BF16W_PB); }
template <int NPIECE> constexpr int bf16w_smem_dw() {
  return 2 * bf16w_oper<NPIECE>() > BF16W_KG * BF16W_BM * BF16W_BN ? 2 * bf16w_oper<NPIECE>() : BF16W_KG * BF16W_BM * BF16W_BN;
}
static_assert(bf16w_oper<3>() == BF16W_OPER && bf16w_smem_dw<3>() == 2 * BF16W_OPER && bf16w_smem_dw<1>() == BF16W_KG * BF16W_BM * BF16W_BN, "plane buffers");

// NPIECE = 3: the exact three-way split (fp32-accurate).  NPIECE = 1 (round 6): every operand rounded to ONE bf16, or with F16 to one IEEE fp16 after
// its power-of-two scale (GemmP::a_scale / b_scale, undone on the accumulator in fp32) -- the arithmetic of gemm_bf16k's one-piece loop: one MFMA per
// wave and K-tile, one plane per operand (48 KiB of planes; the workgroup's 64 KiB are the epilogue's), same staging, prefetch and waits.
template <bool BKM, int NPIECE = 3, bool F16 = false>
__device__ __forceinline__ void bf16w_mainloop(const GemmP& p, const int tm, const int tn, const int bz, float* __restrict__ smem, f32x16& acc_out) {
  static_assert(NPIECE == 3 || NPIECE == 1, "pieces per operand");
  static_assert(!F16 || NPIECE == 1, "fp16 pieces only in the single-piece mode");
  constexpr int BM = BF16W_BM, BN = BF16W_BN, BK = BF16W_BK, PD = BF16W_PD, PA = BF16W_PA, PB = BF16W_PB, OPER = bf16w_oper<NPIECE>();
  const float sa = (F16 && p.a_scale != 0.f) ? p.a_scale : 1.f, sb = (F16 && p.b_scale != 0.f) ? p.b_scale : 1.f;
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  unsigned* const buf = reinterpret_cast<unsigned*>(smem);      // [2][A planes x 3 | B planes x 3]
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kg = wave >> 1, wr = wave & 1;
  const int li = lane & 31, lh = lane >> 5;
  const int m0 = tm * BM, n0 = tn * BN;
  const int nt = (p.K + BK - 1) / BK;

  f32x16 acc, accl;
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc[r] = 0.f; accl[r] = 0.f; }

  // this thread's items: eight consecutive k (k-octet c8) of A row m0 + arow -- two 16-byte loads, one 16-byte store per piece -- and the
  // k-quad kq of B row n0 + row (one load, one 8-byte store per piece)
  const int row = tid >> 5, kq = tid & 31;
  const int arow = tid >> 4, c8 = tid & 15;
  const float* const zp = p.zero_page + (tid & 255) * 8;
  const bool ok0 = m0 + arow < p.M, ok2 = n0 + row < p.N;
  const float* s0 = ok0 ? p.A + (size_t)bz * p.a_batch_stride + (size_t)(m0 + arow) * p.lda + 8 * c8 : zp;
  // B K-major ([K, ldb], n contiguous): the item is four consecutive n of ONE k-row -- k-row tid / 8 of the K-tile, n-quad tid % 8 (eight
  // threads along a k-row's 128 bytes); k-rows past K do not exist (zero page), columns N .. ldb - 1 are zero
  const int kb = tid >> 3, nq = tid & 7;
  const bool okn = n0 + 4 * nq < p.ldb;
  const float* s2 = !BKM ? (ok2 ? p.B + (size_t)(n0 + row) * p.ldb + 4 * kq : zp) : (okn ? p.B + (size_t)kb * p.ldb + n0 + 4 * nq : zp);
  const int i0 = ok0 ? BK : 0;
  const long long i2 = !BKM ? (ok2 ? BK : 0) : (okn ? (long long)BK * p.ldb : 0);
  int kleft = p.K;      // (K-major B) k-rows from the next tile to request to the end of the range
  // columns K .. ld - 1 of an operand row are zero (every leading dimension is a multiple of 64 floats and pads are never stored to:
  // what the 64-deep K-tiles of the other kernels rely on); the LAST 128-deep tile may reach past ld, where the next row begins
  const bool in_a = (nt - 1) * BK + 8 * c8 < p.lda, in_b = (nt - 1) * BK + 4 * kq < p.ldb;
  const int dsta = arow * (BK / 2) + ((c8 ^ (arow & 15)) << 2);                                // dword offsets inside a plane
  const int dst = row * (BK / 2) + ((((kq >> 1) ^ (row & 15))) << 2) + ((kq & 1) << 1);
  // K-major B image: [k][32 n] bf16, 64 bytes per k-row, no XOR -- a half-wave's stores and its transposing reads each cover four
  // consecutive k-rows = 256 contiguous bytes
  const int dstb = !BKM ? dst : kb * (BN / 2) + 2 * nq;

  f32x4k rg[PD][3];
#pragma unroll
  for (int s = 0; s < PD; ++s) { rg[s][0] = f32x4k{0.f, 0.f, 0.f, 0.f}; rg[s][1] = rg[s][0]; rg[s][2] = rg[s][0]; }
  int tload = 0;      // next K-tile to request
  auto load_tile = [&](auto ss) {
    constexpr int s = decltype(ss)::value;
    const bool last = tload == nt - 1, past = tload >= nt;      // (past: a prologue slot behind a short K range -- requested all the same, from the zero page)
    const float* a0 = ((last && !in_a) || past) ? zp : s0;
    const float* a1 = a0 + 4;
    const float* b0 = !BKM ? (((last && !in_b) || past) ? zp : s2) : ((kb < kleft) ? s2 : zp);
    kleft -= BK;
    f32x4k l0, l1, l2;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(l0) : "v"(a0) : "memory");
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(l1) : "v"(a1) : "memory");
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(l2) : "v"(b0) : "memory");
    rg[s][0] = l0; rg[s][1] = l1; rg[s][2] = l2;
    s0 += i0; s2 += i2;
    ++tload;
  };
  // `nwait` = vector-memory loads issued after the three of the K-tile in slot s (loads return in order)
  auto store_tile = [&](auto ss, int b, int nwait) {
    constexpr int s = decltype(ss)::value;
    const f32x4k v0 = rg[s][0], v1 = rg[s][1], v2 = rg[s][2];
    const int tok = bf16w_wait_vm(nwait, v0, v1, v2);
    unsigned* o = buf + b * OPER + dsta;
    unsigned* ob = buf + b * OPER + NPIECE * PA + dstb;
    if constexpr (NPIECE == 3) {
      unsigned h0, m0_, l0, h1, m1, l1, h2, m2, l2, h3, m3, l3;
      split_bf16x3_tok(v0.x, v0.y, h0, m0_, l0, tok); split_bf16x3_tok(v0.z, v0.w, h1, m1, l1, tok);
      split_bf16x3_tok(v1.x, v1.y, h2, m2, l2, tok); split_bf16x3_tok(v1.z, v1.w, h3, m3, l3, tok);
      *reinterpret_cast<u32x4*>(o) = u32x4{h0, h1, h2, h3};
      *reinterpret_cast<u32x4*>(o + PA) = u32x4{m0_, m1, m2, m3};
      *reinterpret_cast<u32x4*>(o + 2 * PA) = u32x4{l0, l1, l2, l3};
      split_bf16x3_tok(v2.x, v2.y, h0, m0_, l0, tok); split_bf16x3_tok(v2.z, v2.w, h1, m1, l1, tok);
      *reinterpret_cast<u32x2*>(ob) = u32x2{h0, h1};
      *reinterpret_cast<u32x2*>(ob + PB) = u32x2{m0_, m1};
      *reinterpret_cast<u32x2*>(ob + 2 * PB) = u32x2{l0, l1};
    } else if constexpr (F16) {      // (every element is multiplied by its operand's scale before anything else reads it: the scale carries the token)
      const float ca = __uint_as_float(__float_as_uint(sa) | (unsigned)tok), cb = __uint_as_float(__float_as_uint(sb) | (unsigned)tok);
      *reinterpret_cast<u32x4*>(o) = u32x4{cvt_pk_f16(v0.x * ca, v0.y * ca), cvt_pk_f16(v0.z * ca, v0.w * ca),
                                           cvt_pk_f16(v1.x * ca, v1.y * ca), cvt_pk_f16(v1.z * ca, v1.w * ca)};
      *reinterpret_cast<u32x2*>(ob) = u32x2{cvt_pk_f16(v2.x * cb, v2.y * cb), cvt_pk_f16(v2.z * cb, v2.w * cb)};
    } else {
      *reinterpret_cast<u32x4*>(o) = u32x4{cvt_pk_bf16_tok(v0.x, v0.y, tok), cvt_pk_bf16_tok(v0.z, v0.w, tok),
                                           cvt_pk_bf16_tok(v1.x, v1.y, tok), cvt_pk_bf16_tok(v1.z, v1.w, tok)};
      *reinterpret_cast<u32x2*>(ob) = u32x2{cvt_pk_bf16_tok(v2.x, v2.y, tok), cvt_pk_bf16_tok(v2.z, v2.w, tok)};
    }
  };
  // prologue: K-tiles 0 .. PD-1 requested, tile 0 split into buffer 0; everything requested lands before the loop is entered
  // (all PD slots without a branch: with the requests inside `if (slot < nt)` hipcc joins the paths with COPIES of registers whose loads
  // are in flight -- tools/check_asm_prefetch.py)
  static_for<0, PD>([&](auto ss) { load_tile(ss); });
  store_tile(std::integral_constant<int, 0>{}, 0, 3 * (PD - 1));
  asm volatile("s_waitcnt vmcnt(0)" :: "v"(rg[0][0]), "v"(rg[0][1]), "v"(rg[0][2]), "v"(rg[1][0]), "v"(rg[1][1]), "v"(rg[1][2]));
  static_assert(PD == 2, "the wait above names every prefetch register");
  __syncthreads();
#ifdef GANMF_PERSIST_DIAG_BUILD
  if (p.stamps && tid == 0) p.stamps[(size_t)blockIdx.x * 4 + 1] = __builtin_amdgcn_s_memrealtime();      // first K-tile split into LDS
#endif

  const int fa = (wr * 32 + li) * (BK / 2) + (((2 * kg + lh) ^ (li & 15)) << 2);      // this lane's fragment of an A plane / of a B plane
  const int fb = li * (BK / 2) + (((2 * kg + lh) ^ (li & 15)) << 2);
  // K-major B: ds_read_b64_tr_b16 (SplitStage<.., true>::frag): lane 4 qq + pp of a 16-lane group gives the address of k-row qq, columns
  // 4 pp .. 4 pp + 3 of a 4 (k) x 16 (n) block and receives column (lane & 15) of the four k-rows; group li >> 4 takes columns 16 .. 31;
  // lane half lh the k-rows 16 kg + 8 lh + 0..7 in two reads (+ 0..3, + 4..7)
  const int fbt = ((16 * kg + 8 * lh + ((li & 15) >> 2)) * (BN / 2) * 4) + 32 * (li >> 4) + 8 * (li & 3);      // bytes inside a B plane
  auto step = [&](auto ss, int t) {      // K-tile t, whose registers were slot s = t % PD (consumed in iteration t - 1)
    constexpr int s = decltype(ss)::value, s1_ = (s + 1) % PD;
    const unsigned* planes = buf + (t & 1) * OPER;
    u32x4 pa[3], pb[3];
#pragma unroll
    for (int q = 0; q < NPIECE; ++q) pa[q] = *reinterpret_cast<const u32x4*>(planes + q * PA + fa);
    if constexpr (!BKM) {
#pragma unroll
      for (int q = 0; q < NPIECE; ++q) pb[q] = *reinterpret_cast<const u32x4*>(planes + NPIECE * PA + q * PB + fb);
    } else {
#if defined(__HIP_DEVICE_COMPILE__)      // (LDS pointers are 32 bits wide in the device pass only)
      typedef __attribute__((address_space(3))) s16x4* lds_s16x4_p;
      const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) const unsigned*)(planes + NPIECE * PA) + (unsigned)fbt;
#pragma unroll
      for (int q = 0; q < NPIECE; ++q) {
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(base + q * PB * 4));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(base + q * PB * 4 + 4 * (BN / 2) * 4));
        pb[q] = __builtin_bit_cast(u32x4, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
      }
#else
      (void)fbt;
#pragma unroll
      for (int q = 0; q < NPIECE; ++q) pb[q] = u32x4{0u, 0u, 0u, 0u};
#endif
    }
    if (t + PD < nt) load_tile(ss);
    constexpr int ta[6] = {1, 0, 2, 1, 0, 0}, tb[6] = {1, 2, 0, 0, 1, 0};   // (mid,mid) (hi,lo) (lo,hi) (mid,hi) (hi,mid) (hi,hi): gemm_bf16s_body
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
    // under the MFMAs: split K-tile t + 1 into the other buffer; behind its three loads went those of tiles t + 2 .. min(t + PD, nt - 1)
    if (t + 1 < nt) store_tile(std::integral_constant<int, s1_>{}, (t + 1) & 1, 3 * (min(t + PD, nt - 1) - (t + 1)));
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

template <bool BKM, int NPIECE = 3, bool F16 = false>
__device__ __forceinline__ void gemm_bf16w_body(const GemmP& p, const int bid, const int nblk, float* __restrict__ smem) {
  int tm, tn, sp, bz;
  tile_coords(p, bid, nblk, tm, tn, sp, bz);
  f32x16 acc[1][1];
#ifdef GANMF_PERSIST_DIAG_BUILD
  if (p.stamps && threadIdx.x == 0) p.stamps[(size_t)blockIdx.x * 4 + 0] = __builtin_amdgcn_s_memrealtime();
#endif
  bf16w_mainloop<BKM, NPIECE, F16>(p, tm, tn, bz, smem, acc[0][0]);
#ifdef GANMF_PERSIST_DIAG_BUILD
  if (p.stamps && threadIdx.x == 0) p.stamps[(size_t)blockIdx.x * 4 + 2] = __builtin_amdgcn_s_memrealtime();
#endif
  // the eight K groups' 64 x 32 images (the loop's last barrier is behind every fragment read)
  static_assert(BF16W_KG * BF16W_BM * BF16W_BN <= bf16w_smem_dw<NPIECE>(), "the workgroup's LDS must hold the staged partial tiles");
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int kg = wave >> 1, wr = wave & 1, li = lane & 31, lh = lane >> 5;
  float* ct = smem + kg * (BF16W_BM * BF16W_BN);
#pragma unroll
  for (int r = 0; r < 16; ++r) ct[(wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * BF16W_BN + li] = acc[0][0][r];
  gemm_epilogue<BF16W_BM, BF16W_BN, 1, 1, 4, false, BF16W_KG, true>(p, acc, smem, TileCoord{tm, tn, sp, bz, tm * BF16W_BM, tn * BF16W_BN});
#ifdef GANMF_PERSIST_DIAG_BUILD
  if (p.stamps) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); if (threadIdx.x == 0) p.stamps[(size_t)blockIdx.x * 4 + 3] = __builtin_amdgcn_s_memrealtime(); }
#endif
}

template <bool BKM, int NPIECE = 3, bool F16 = false>
__global__ __launch_bounds__(1024) void gemm_bf16w_mfma(const GemmP p) {
  __shared__ __attribute__((aligned(16))) float smem[bf16w_smem_dw<NPIECE>()];      // three pieces: 144 KiB, one workgroup per CU; one piece: 64 KiB
  gemm_bf16w_body<BKM, NPIECE, F16>(p, (int)blockIdx.x, (int)gridDim.x, smem);
}

// NT / NN products of plain fp32 matrices with nothing in the fetch path that the 64 x 64 kernels' items carry
inline bool gemm_bf16w_eligible(const GemmP& p, bool akm, bool bkm) {
  // leading dimensions: the kernel issues 16-byte loads at A + row * lda + 8 c and B + row * ldb + 4 c, and its K-tail logic reads up to the next
  // multiple of 64 past K inside a row (zero pads, as every matrix of the library has them, handle.inc LD_ALIGN) -- anything else stays on 64 x 64
  return !akm && std::max(p.nbatch, 1) == 1 && !p.a_gather && !p.a_planes && !p.b_planes && !p.epi.csr_indptr && !p.epi.sp_rows &&
         p.epi.kind != EPI_ADAM && p.K >= 1 && (p.lda % 64) == 0 && (p.ldb % 64) == 0;      // (the zero page is checked at dispatch: plans are drawn before it is filled in)
}
// out-of-range rows read zero_page + (tid & 255) * 8 + 0..7: the page must hold at least this many floats (the handle's: 2048 + 64, abi_core.inc)
constexpr int BF16W_ZERO_PAGE_FLOATS = 255 * 8 + 8;
static_assert(BF16W_ZERO_PAGE_FLOATS <= 2048 + 64, "gemm_bf16w reads past the library's zero page");

// `force`: whenever eligible (tests); otherwise where the 64 x 64 plan spreads a few tiles over the chip as K slices + a slab-sum launch
// and the 64 x 32 grid fits the chip in one round, by the two kernels' measured costs (one workgroup per CU):
// 64 x 64: 5.3 us + 0.94 us per 64-deep K-tile of a slice + 4.7 us for the slab sum; 64 x 32: 4.4 us + 1.3 us per 128-deep K-tile
inline bool plan_bf16w(GemmPlan& pl, const GemmP& g, bool akm, bool bkm, bool force) {
  if (!gemm_bf16w_eligible(g, akm, bkm)) return false;
  const int tm = (g.M + BF16W_BM - 1) / BF16W_BM, tn = (g.N + BF16W_BN - 1) / BF16W_BN;
  // one low-precision piece per operand (a forced bf16 / fp16 plan on the 16-wave kernel): the same tiles and K-tiles, one MFMA per wave and K-tile --
  // measured on MI355X (round 6, tools/k_sweep.py): 64 x 64 about 0.45 us per 64-deep K-tile (fwd of configs[4]: 11.8 us for 14.5 K-tiles), 64 x 32 about 0.75 us per 128-deep K-tile (dF: 10.5 us for 8)
  const bool lp = pl.mode == MFMA_F16 || pl.mode == MFMA_BF16;
  const double k64 = lp ? 0.45 : 0.94, k32 = lp ? 0.75 : 1.3;
  if (!force) {
    if (!((pl.mode == MFMA_BF16X3 || lp) && pl.tile == 64 && pl.kg == 4 && pl.nsplit >= 2 && !pl.persist && !pl.skinny && !pl.skinny_n)) return false;
    if ((long long)tm * tn > GEMM_CUS) return false;
    const double est64 = 5.3 + k64 * ((pl.kps + 63) / 64) + 4.7, est32 = 4.4 + k32 * ((g.K + BF16W_BK - 1) / BF16W_BK);
    if (est32 > est64 - 0.5) return false;
  }
  const bool wants_sq = pl.sq_count > 0 || g.epi.sq_partials != nullptr;
  pl.est_us = 4.4 + k32 * ((g.K + BF16W_BK - 1) / BF16W_BK);
  pl.wide32 = 1; pl.skinny = 0; pl.skinny_n = 0; pl.persist = 0;
  if (!lp) pl.mode = MFMA_BF16X3;
  pl.tile = 64; pl.kg = BF16W_KG; pl.ring = 2; pl.bk = BF16W_BK;
  pl.tiles_m = tm; pl.tiles_n = tn; pl.nsplit = 1; pl.kps = g.K;
  pl.sq_count = wants_sq ? tm * tn : 0;
  return true;
}

inline hipError_t gemm_dispatch_bf16w(hipStream_t st, const GemmP& p0, bool bkm, int mode) {      // (tiles / split of the plan: gemm_run)
  GemmP p = p0;
  const int grid = p.tiles_m * p.tiles_n * std::max(p.nbatch, 1);
  if (grid <= 0) return hipSuccess;
#ifdef GANMF_PERSIST_DIAG_BUILD
  const bool stamping = gemm_stamps_on() && gemm_stamps_begin(p, grid, st);
  struct Report { const GemmP& p; int grid; hipStream_t st; bool bkm, on; ~Report() { if (on) gemm_stamps_report(p, grid, st, false, bkm, "gemm_bf16w_mfma (64 x 32 tiles, 128-deep K-tiles)"); } } report{p, grid, st, bkm, stamping};
#endif
  if (p.nsplit != 1 || p.tiles_m != (p.M + BF16W_BM - 1) / BF16W_BM || p.tiles_n != (p.N + BF16W_BN - 1) / BF16W_BN) return hipErrorInvalidValue;
  if (!p.zero_page || (p.lda % 64) != 0 || (p.ldb % 64) != 0) return hipErrorInvalidValue;      // (gemm_bf16w_eligible: never a silent out-of-bounds read)
  if (mode == MFMA_F16) {
    if (bkm) GANMF_LAUNCH((gemm_bf16w_mfma<true, 1, true>), dim3(grid), dim3(1024), 0, st, p);
    else GANMF_LAUNCH((gemm_bf16w_mfma<false, 1, true>), dim3(grid), dim3(1024), 0, st, p);
  } else if (mode == MFMA_BF16) {
    if (bkm) GANMF_LAUNCH((gemm_bf16w_mfma<true, 1, false>), dim3(grid), dim3(1024), 0, st, p);
    else GANMF_LAUNCH((gemm_bf16w_mfma<false, 1, false>), dim3(grid), dim3(1024), 0, st, p);
  } else {
    if (bkm) GANMF_LAUNCH(gemm_bf16w_mfma<true>, dim3(grid), dim3(1024), 0, st, p);
    else GANMF_LAUNCH(gemm_bf16w_mfma<false>, dim3(grid), dim3(1024), 0, st, p);
  }
  return hipGetLastError();
}

}  // namespace ganmf
