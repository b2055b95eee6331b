// Skinny-K products: C[M, N] = A[M, K] . op(B) with K <= 64 and a wide N -- the decode and dF products of a narrow autoencoder
// (emb_dim 32: the reference's default, GANMF.py:146, and BASELINE configs[3] at e = 32), the generator product of a small
// num_factors.  At K = 33 a [256, 50 000] output is 51 MB written (+ 51 MB of the subtracted input read) against 0.84 GFLOP: the
// product is a STREAM with a little arithmetic on the way, and the tiled MFMA kernels -- built around a long K loop, with an epilogue
// that keeps two 16-byte accesses per thread in flight -- move it at 1.7-2.5 TB/s (decode 44 us, dF 25 us at configs[3] e = 32).
// Here: one workgroup = 64 rows x BN columns; A tile and B strip sit in LDS (45 KiB at K = 33, BN = 256: three workgroups per
// CU); a thread owns RT rows x 4 consecutive columns, requests ALL its epilogue operands (RT float4 of `aux`) before the K loop,
// accumulates in fp32 FMAs with k ascending (plain fp32 arithmetic: no split, no rounding of the operands) and stores 16 bytes per
// row -- a wave writes whole 256-byte .. 1-KiB row segments.  Epilogues: store, `- aux` with the sum of squares (decode), `- c * aux`
// (dF).  Batches as in GemmP.  Deterministic: fixed k order, one partial sum of squares per workgroup, filed by tile.
#pragma once
#include "kernels.hpp"

namespace ganmf {

constexpr int SKINNY_KMAX = 64;
constexpr int SKINNY_BN = 256;      // one wave = RT rows x 256 columns (64 lanes x 4)
typedef float sk_f4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(4))) const sk_f4 sk_cf4;      // A is never written by the kernel that reads it this way

// RT rows per wave: the workgroup's tile is 4 * RT rows x 256 columns.  The B strip [K][256] sits in LDS (36 KiB at K = 33: four
// workgroups per CU).  The A values a wave needs are the same for all its lanes -- row r, four consecutive k -- so they are read
// through the SCALAR cache straight from global memory (s_load_dwordx4, SGPR operands of v_pk_fma_f32): with A in LDS every FMA pair
// cost as much LDS time as arithmetic (a first version of this kernel: 35 us for the [256, 50 000] x K = 33 product, all of it LDS
// issue, 64 B/clk per CU for 16-byte reads); now LDS carries four 16-byte reads per lane per four k and 16 rows.
template <int RT, bool BKM>
__global__ __launch_bounds__(256) void gemm_skinny_kernel(const GemmP p) {
  constexpr int BN = SKINNY_BN, BM = 4 * RT, C4 = BN / 4;
  extern __shared__ __attribute__((aligned(16))) float sk_smem[];
  float* __restrict__ sB = sk_smem;                  // [Kp][BN]
  const int Kp = (p.K + 3) & ~3;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int per = p.tiles_m * p.tiles_n;
  const int bz = (int)blockIdx.x / per, rem = (int)blockIdx.x % per;
  const int tm = rem % p.tiles_m, tn = rem / p.tiles_m;      // tile row fastest: the workgroups of one B strip run next to each other
  const int m0 = tm * BM, n0 = tn * BN;
  const float* __restrict__ A = p.A + (size_t)bz * p.a_batch_stride;
  float* __restrict__ C = p.C + (size_t)bz * p.c_batch_stride;
  const EpiD& e = p.epi;
  const bool want_aux = e.kind == EPI_SUB_AUX_SQ || e.kind == EPI_SUB_SCALED_AUX;
  const float* __restrict__ aux = want_aux ? e.aux + (size_t)bz * e.aux_batch_stride : nullptr;
  const int col = n0 + 4 * lane;

  // B strip as [k][n]: every thread first REQUESTS all its pieces, then files them (a "load, store, next" loop with a run-time trip
  // count sends the loads out one memory latency at a time)
  const int kq = Kp / 4;
  constexpr int IT_B = BKM ? (SKINNY_KMAX * C4 + 255) / 256 : (BN * (SKINNY_KMAX / 4) + 255) / 256;
  {
    float4 vb[IT_B];
#pragma unroll
    for (int it = 0; it < IT_B; ++it) {
      const int idx = tid + it * 256;
      vb[it] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (BKM) {      // B is [K, ldb] row-major: rows of the strip are contiguous
        const int k = idx / C4, c = idx % C4;
        if (idx < Kp * C4 && k < p.K && n0 + 4 * c + 3 < p.ldb) vb[it] = *reinterpret_cast<const float4*>(p.B + (size_t)k * p.ldb + n0 + 4 * c);
      } else {        // B is [N, ldb], K-contiguous: element (k, n) = B[n * ldb + k]; consecutive lanes take consecutive rows n
        const int n = idx % BN, q = idx / BN;
        if (idx < BN * kq && n0 + n < p.N) {
          const float* s = p.B + (size_t)(n0 + n) * p.ldb + 4 * q;
          if (4 * q + 3 < p.K) vb[it] = *reinterpret_cast<const float4*>(s);
          else { if (4 * q < p.K) vb[it].x = s[0]; if (4 * q + 1 < p.K) vb[it].y = s[1]; if (4 * q + 2 < p.K) vb[it].z = s[2]; }
        }
      }
    }
#pragma unroll
    for (int it = 0; it < IT_B; ++it) {
      const int idx = tid + it * 256;
      if (BKM) {
        if (idx < Kp * C4) *reinterpret_cast<float4*>(sB + (idx / C4) * BN + 4 * (idx % C4)) = vb[it];
      } else if (idx < BN * kq) {      // conflict-free writes (lanes along n)
        const int n = idx % BN, q = idx / BN;
        sB[(4 * q + 0) * BN + n] = vb[it].x; sB[(4 * q + 1) * BN + n] = vb[it].y; sB[(4 * q + 2) * BN + n] = vb[it].z; sB[(4 * q + 3) * BN + n] = vb[it].w;
      }
    }
  }
  __syncthreads();

  // this wave's rows (wave-uniform); rows past M repeat the last one (computed, never stored)
  const int r0 = m0 + wave * RT;
  float4 acc[RT];
#pragma unroll
  for (int i = 0; i < RT; ++i) acc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  const float* __restrict__ bcol = sB + 4 * lane;
  // k >= K inside the last group of four: those rows of the strip are zero, and A's pad columns are zero by the buffer layout every
  // kernel of the library relies on (leading dimensions are whole K-tiles, pads are zero and are never stored to).
  // (Masking A's values there was measured: inside the loop it costs the scalar loads their pipelining, 26.7 -> 31.6 us; as a peeled
  // last group 28.8 us.)
  for (int k4 = 0; k4 < Kp; k4 += 4) {
    const float4 b0 = *reinterpret_cast<const float4*>(bcol + (k4 + 0) * BN);
    const float4 b1 = *reinterpret_cast<const float4*>(bcol + (k4 + 1) * BN);
    const float4 b2 = *reinterpret_cast<const float4*>(bcol + (k4 + 2) * BN);
    const float4 b3 = *reinterpret_cast<const float4*>(bcol + (k4 + 3) * BN);
#pragma unroll
    for (int i = 0; i < RT; ++i) {
      const int row = min(r0 + i, p.M - 1);
      // uniform address, read through the CONSTANT address space: a scalar load (s_load_dwordx4) into SGPRs that v_pk_fma_f32 takes as
      // an operand
      const sk_f4 a = *reinterpret_cast<const sk_cf4*>(reinterpret_cast<unsigned long long>(A + (size_t)row * p.lda + k4));
      acc[i].x = fmaf(a.x, b0.x, acc[i].x); acc[i].y = fmaf(a.x, b0.y, acc[i].y); acc[i].z = fmaf(a.x, b0.z, acc[i].z); acc[i].w = fmaf(a.x, b0.w, acc[i].w);
      acc[i].x = fmaf(a.y, b1.x, acc[i].x); acc[i].y = fmaf(a.y, b1.y, acc[i].y); acc[i].z = fmaf(a.y, b1.z, acc[i].z); acc[i].w = fmaf(a.y, b1.w, acc[i].w);
      acc[i].x = fmaf(a.z, b2.x, acc[i].x); acc[i].y = fmaf(a.z, b2.y, acc[i].y); acc[i].z = fmaf(a.z, b2.z, acc[i].z); acc[i].w = fmaf(a.z, b2.w, acc[i].w);
      acc[i].x = fmaf(a.w, b3.x, acc[i].x); acc[i].y = fmaf(a.w, b3.y, acc[i].y); acc[i].z = fmaf(a.w, b3.z, acc[i].z); acc[i].w = fmaf(a.w, b3.w, acc[i].w);
    }
  }

  // epilogue in chunks of EC rows: the chunk's `aux` operands are requested together, then applied and stored (16 bytes per lane: a
  // wave writes one whole 1-KiB row segment per store instruction).  Measured on the [256, 50 000] product: 15 us with K <= 4 (the
  // output stream), + 0.34 us per k; computing and storing in row chunks, so that stores leave under the next chunk's arithmetic,
  // changed nothing at this size (27 us either way at K = 33) and doubled the time of the small products.
  constexpr int EC = RT < 8 ? RT : 8;
  float sq = 0.f;
#pragma unroll
  for (int ch = 0; ch < RT; ch += EC) {
    float4 ax[EC];
#pragma unroll
    for (int i = 0; i < EC; ++i) {
      const int row = r0 + ch + i;
      ax[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (aux && row < p.M && col < p.N) {
        const float* s = aux + (size_t)row * e.ldaux + col;
        if (col + 3 < p.N) ax[i] = *reinterpret_cast<const float4*>(s);
        else { ax[i].x = s[0]; if (col + 1 < p.N) ax[i].y = s[1]; if (col + 2 < p.N) ax[i].z = s[2]; }
      }
    }
#pragma unroll
    for (int i = 0; i < EC; ++i) {
      const int row = r0 + ch + i;
      if (row < p.M && col < p.N) {
        float4 v = acc[ch + i];
        const float4 x = ax[i];
        if (e.kind == EPI_SUB_AUX_SQ) {
          v.x -= x.x; v.y -= x.y; v.z -= x.z; v.w -= x.w;
          sq += v.x * v.x;
          if (col + 1 < p.N) sq += v.y * v.y;
          if (col + 2 < p.N) sq += v.z * v.z;
          if (col + 3 < p.N) sq += v.w * v.w;
        } else if (e.kind == EPI_SUB_SCALED_AUX) {
          v.x -= e.c * x.x; v.y -= e.c * x.y; v.z -= e.c * x.z; v.w -= e.c * x.w;
        }
        float* d = C + (size_t)row * p.ldc + col;
        if (col + 3 < p.N) *reinterpret_cast<float4*>(d) = v;
        else { d[0] = v.x; if (col + 1 < p.N) d[1] = v.y; if (col + 2 < p.N) d[2] = v.z; }
      }
    }
  }
  if (e.sq_partials) {
    sq = wave_sum(sq);
    __syncthreads();      // the strip is no longer read: its first words carry the four wave sums
    if (lane == 0) sB[wave] = sq;
    __syncthreads();
    if (tid == 0) e.sq_partials[(size_t)bz * e.sq_stride + tn * p.tiles_m + tm] = (sB[0] + sB[1]) + (sB[2] + sB[3]);
  }
}

// Which products take this kernel: plain fp32 arithmetic is at least what every MFMA mode promises, so any handle arithmetic may
// use it; the shapes are those where the output stream dominates (K <= 64, N >= 2048) and the epilogue is one it implements.
inline bool gemm_skinny_eligible(const GemmP& p, bool akm) {
  const int kind = p.epi.kind;
  return !akm && p.K >= 1 && p.K <= SKINNY_KMAX && p.N >= 2048 && p.a_gather == nullptr && p.epi.csr_indptr == nullptr &&
         p.epi.sp_rows == nullptr && p.epi.sq_m_half == 0 && (p.lda % 4) == 0 && p.lda >= ((p.K + 3) & ~3) && (p.ldb % 4) == 0 && (p.ldc % 4) == 0 &&
         (kind == EPI_STORE || ((kind == EPI_SUB_AUX_SQ || kind == EPI_SUB_SCALED_AUX) && p.epi.aux != nullptr && (p.epi.ldaux % 4) == 0));
}
// rows per wave: 16 (tiles of 64 rows), or 8 / 4 when that leaves fewer than two workgroups per CU
inline int gemm_skinny_rt(int M, int N, int nbatch) {
  const long long tn = (N + SKINNY_BN - 1) / SKINNY_BN;
  for (int rt : {16, 8}) {
    if ((long long)((M + 4 * rt - 1) / (4 * rt)) * tn * std::max(nbatch, 1) >= 2LL * GEMM_CUS) return rt;
  }
  return 4;
}
// the plan of a product that takes this kernel (unsplit, one partial sum of squares per workgroup)
inline bool plan_skinny(GemmPlan& pl, const GemmP& g, bool akm) {
  if (!gemm_skinny_eligible(g, akm)) return false;
  pl.skinny = gemm_skinny_rt(g.M, g.N, g.nbatch);
  pl.nsplit = 1; pl.persist = 0; pl.kps = (g.K + 3) & ~3;
  pl.tiles_m = (g.M + 4 * pl.skinny - 1) / (4 * pl.skinny);
  pl.tiles_n = (g.N + SKINNY_BN - 1) / SKINNY_BN;
  pl.sq_count = pl.tiles_m * pl.tiles_n;
  return true;
}

inline hipError_t gemm_dispatch_skinny(hipStream_t st, const GemmP& p0, bool bkm, int rt) {
  GemmP p = p0;
  if (p.nbatch < 1) p.nbatch = 1;
  p.tiles_m = (p.M + 4 * rt - 1) / (4 * rt);
  p.tiles_n = (p.N + SKINNY_BN - 1) / SKINNY_BN;
  const int Kp = (p.K + 3) & ~3;
  const size_t lds = (size_t)SKINNY_BN * Kp * sizeof(float);      // <= 64 KiB at K = 64
  const int grid = p.tiles_m * p.tiles_n * p.nbatch;
  if (grid <= 0) return hipSuccess;
#define GANMF_SKINNY(RT_)                                                                                         \
  do {                                                                                                            \
    if (bkm) GANMF_LAUNCH((gemm_skinny_kernel<RT_, true>), dim3(grid), dim3(256), lds, st, p);                    \
    else GANMF_LAUNCH((gemm_skinny_kernel<RT_, false>), dim3(grid), dim3(256), lds, st, p);                       \
  } while (0)
  if (rt == 16) GANMF_SKINNY(16);
  else if (rt == 8) GANMF_SKINNY(8);
  else GANMF_SKINNY(4);
#undef GANMF_SKINNY
  return hipGetLastError();
}


// ---- Skinny-N products (round 5): C[M, N <= 32] = A[M, K] . op(B) with a LONG K ---------------------------------------------------------------
// The encode product and dE of a narrow autoencoder (emb_dim 32: the reference's default, GANMF.py:88, and BASELINE configs[3] at e = 32) read a
// [2B, 50 000] activation matrix (51 MB) for a [2B, 32] result: every A element is used ONCE, the product is a stream over A with 16 FLOP per
// byte.  The tiled kernels give such a product 64 x 64 tiles of which half the columns are padding, stage A through LDS and walk K in 64-wide
// tiles behind a barrier each: 2.3-3.0 TB/s (encode 19.5 us, dE + d_coef 21.4 us at the configs[3] shard).  Here a workgroup of eight waves owns
// ALL rows of a 256-row block and one 256-wide K slice; wave w = the 32 x 32 output block of rows 32 w .. 32 w + 31:
//   * A goes from global memory STRAIGHT into MFMA operand registers: lane (i, h) loads the 32 bytes k = 16 s + 8 h .. + 7 of row i for step s
//     -- exactly its fragment of v_mfma_f32_32x32x16_bf16 -- and splits them (the exact three-way split of gemm_bf16s.hpp, once per element:
//     nobody else reads them); the slice is fully unrolled, so the sixteen loads a lane keeps in flight are counted exactly by the compiler;
//   * the B slice [256][32] is split ONCE per workgroup into three bf16 planes [n][256 k] in LDS (row stride 264: conflict-free ds_read_b128);
//   * the same six piece products in the same order into the same two accumulators as every split-bf16 kernel of the library (fp32-accurate).
//     (A first version ran the fp32 MFMA on the raw operands: 128 dependent 64-cycle MFMAs per wave = 8 us of matrix pipe per CU, 19 us per
//     launch against 23 for the tiled kernel; the split form needs 3.6 us of SIMD time.)
// Output: one fp32 slab per K slice, summed by splitk_reduce_kernel with the product's epilogue as for every split product (fixed order).
constexpr int SKN_KPS = 256;          // K slice of a workgroup (GemmPlan::kps)
constexpr int SKN_STEPS = SKN_KPS / 16;
constexpr int SKN_PF = 8;             // steps (two 16-byte loads per lane each) in flight
constexpr int SKN_NMAX = 32;
constexpr int SKN_BM = 256;
constexpr int SKN_LDP = SKN_KPS / 2 + 4;      // dwords per plane row: 256 bf16 + 16 bytes
constexpr int SKN_PLANE = 32 * SKN_LDP;       // dwords per piece plane

template <bool BKM>
__global__ __launch_bounds__(512) void gemm_skinny_n_kernel(const GemmP p) {
  __shared__ __attribute__((aligned(16))) unsigned skn_planes[3 * SKN_PLANE];      // 49.5 KiB
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int sp = (int)blockIdx.x % p.nsplit, tm = (int)blockIdx.x / p.nsplit;
  const int kbeg = sp * p.k_per_split;

  // ---- this lane's A stream: requested first, SKN_PF steps deep
  const int row = tm * SKN_BM + wave * 32 + li;
  const bool row_ok = row < p.M;
  const float* __restrict__ arow = p.A + (size_t)(row_ok ? row : 0) * p.lda;
  auto a_src = [&](int s, int half) -> const float* {      // 16 bytes: k = kbeg + 16 s + 8 lh + 4 half .. + 3 (pad columns up to lda are zero; past lda: the zero page)
    const int k = kbeg + 16 * s + 8 * lh + 4 * half;
    return (row_ok && k + 3 < p.lda) ? arow + k : p.zero_page + 4 * (lane & 7);
  };
  sk_f4 ra[SKN_PF][2];
#pragma unroll
  for (int s = 0; s < SKN_PF; ++s) {
    ra[s][0] = *reinterpret_cast<const sk_f4*>(a_src(s, 0));
    ra[s][1] = *reinterpret_cast<const sk_f4*>(a_src(s, 1));
  }

  // ---- B slice -> three bf16 planes [n][k] in LDS (elements past K or N are zero: A's zero padding then meets zeros, never another tensor's
  // bytes).  One item = row n, eight consecutive k: 1024 items, two per thread.
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int idx = tid + it * 512;
    int n, q;
    float x[8];
    if constexpr (BKM) {      // B is [K, ldb] row-major: consecutive lanes take consecutive columns n (128-byte rows)
      n = idx & 31; q = idx >> 5;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int k = kbeg + 8 * q + j;
        x[j] = (n < p.N && k < p.K) ? p.B[(size_t)k * p.ldb + n] : 0.f;
      }
    } else {                  // B is [N, ldb], K-contiguous: element (k, n) = B[n * ldb + k]; consecutive lanes take consecutive k groups
      q = idx & 31; n = idx >> 5;
      const int k = kbeg + 8 * q;
      const float* src = p.B + (size_t)n * p.ldb + k;
      if (n < p.N && k + 7 < p.K) {
        const float4 v0 = *reinterpret_cast<const float4*>(src), v1 = *reinterpret_cast<const float4*>(src + 4);
        x[0] = v0.x; x[1] = v0.y; x[2] = v0.z; x[3] = v0.w; x[4] = v1.x; x[5] = v1.y; x[6] = v1.z; x[7] = v1.w;
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = (n < p.N && k + j < p.K) ? src[j] : 0.f;
      }
    }
    u32x4 pc[3];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      unsigned hh, mm, ll;
      split_bf16x3(x[2 * j], x[2 * j + 1], hh, mm, ll);
      pc[0][j] = hh; pc[1][j] = mm; pc[2][j] = ll;
    }
#pragma unroll
    for (int pi = 0; pi < 3; ++pi) *reinterpret_cast<u32x4*>(skn_planes + pi * SKN_PLANE + n * SKN_LDP + 4 * q) = pc[pi];
  }
  __syncthreads();

  f32x16 acc, accl;
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc[r] = 0.f; accl[r] = 0.f; }
  const unsigned* __restrict__ bl = skn_planes + li * SKN_LDP + 4 * lh;      // this lane's fragment of step 0, piece 0
#pragma unroll
  for (int s = 0; s < SKN_STEPS; ++s) {
    const sk_f4 a0 = ra[s % SKN_PF][0], a1 = ra[s % SKN_PF][1];
    u32x4 pa[3], pb[3];
#pragma unroll
    for (int pi = 0; pi < 3; ++pi) pb[pi] = *reinterpret_cast<const u32x4*>(bl + pi * SKN_PLANE + 8 * s);
    {
      unsigned hh, mm, ll;
      split_bf16x3(a0.x, a0.y, hh, mm, ll); pa[0][0] = hh; pa[1][0] = mm; pa[2][0] = ll;
      split_bf16x3(a0.z, a0.w, hh, mm, ll); pa[0][1] = hh; pa[1][1] = mm; pa[2][1] = ll;
      split_bf16x3(a1.x, a1.y, hh, mm, ll); pa[0][2] = hh; pa[1][2] = mm; pa[2][2] = ll;
      split_bf16x3(a1.z, a1.w, hh, mm, ll); pa[0][3] = hh; pa[1][3] = mm; pa[2][3] = ll;
    }
    if (s + SKN_PF < SKN_STEPS) {
      ra[s % SKN_PF][0] = *reinterpret_cast<const sk_f4*>(a_src(s + SKN_PF, 0));
      ra[s % SKN_PF][1] = *reinterpret_cast<const sk_f4*>(a_src(s + SKN_PF, 1));
    }
    // (mid,mid) (hi,lo) (lo,hi) (mid,hi) (hi,mid) -> accl, (hi,hi) -> acc: the order of gemm_bf16s_body
    accl = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, pa[1]), __builtin_bit_cast(bf16x8, pb[1]), accl, 0, 0, 0);
    accl = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, pa[0]), __builtin_bit_cast(bf16x8, pb[2]), accl, 0, 0, 0);
    accl = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, pa[2]), __builtin_bit_cast(bf16x8, pb[0]), accl, 0, 0, 0);
    accl = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, pa[1]), __builtin_bit_cast(bf16x8, pb[0]), accl, 0, 0, 0);
    accl = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, pa[0]), __builtin_bit_cast(bf16x8, pb[1]), accl, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, pa[0]), __builtin_bit_cast(bf16x8, pb[0]), acc, 0, 0, 0);
  }
  acc += accl;
  // ---- this slice's slab: register r of lane (i, h) is row 8 (r / 4) + 4 h + r % 4, column i of the wave's block
  float* __restrict__ C = p.C + (size_t)sp * p.c_split_stride;
  if (li < p.N) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int ro = tm * SKN_BM + wave * 32 + 8 * (r >> 2) + 4 * lh + (r & 3);
      if (ro < p.M) C[(size_t)ro * p.ldc + li] = acc[r];
    }
  }
}

// Which products: N <= 32 behind a long K (an A of at least 16 MB) with a K-contiguous A, no batches, nothing that only the tiled kernels' operand fetch
// does (row gather, planes); the epilogue is applied by the slab sum, so every kind the reduce kernel knows is fine.
inline bool gemm_skinny_n_eligible(const GemmP& p, bool akm) {
  return !akm && p.N >= 1 && p.N <= SKN_NMAX && p.K >= 2048 && (long long)p.M * p.K >= (4LL << 20) &&      // (below 16 MB of A the tiled kernels' fewer slabs win: 64 x 17 633 13.5 vs 10.7 us)
         std::max(p.nbatch, 1) == 1 && p.a_gather == nullptr && p.a_planes == nullptr &&
         p.epi.csr_indptr == nullptr && p.epi.sp_rows == nullptr && p.epi.sq_m_half == 0 && p.epi.kind != EPI_ADAM &&
         (p.lda % 4) == 0 && (p.ldb % 4) == 0;
}
inline bool plan_skinny_n(GemmPlan& pl, const GemmP& g, bool akm) {
  if (!gemm_skinny_n_eligible(g, akm)) return false;
  const int ns = (g.K + SKN_KPS - 1) / SKN_KPS;
  if (ns < 2) return false;
  pl.skinny = 0; pl.skinny_n = 1; pl.persist = 0;
  pl.nsplit = ns; pl.kps = SKN_KPS;
  pl.tiles_m = (g.M + SKN_BM - 1) / SKN_BM; pl.tiles_n = 1;
  pl.mode = MFMA_BF16X3; pl.tile = 64; pl.kg = 1;
  pl.sq_count = 0;      // (partials come from the slab sum: run_gemm / gemm_run set their count)
  return true;
}
inline hipError_t gemm_dispatch_skinny_n(hipStream_t st, const GemmP& p0, bool bkm) {
  GemmP p = p0;
  if (p.nsplit < 2 || p.k_per_split != SKN_KPS || !p.zero_page) return hipErrorInvalidValue;
  const int grid = p.nsplit * ((p.M + SKN_BM - 1) / SKN_BM);
  if (bkm) GANMF_LAUNCH((gemm_skinny_n_kernel<true>), dim3(grid), dim3(512), 0, st, p);
  else GANMF_LAUNCH((gemm_skinny_n_kernel<false>), dim3(grid), dim3(512), 0, st, p);
  return hipGetLastError();
}

}  // namespace ganmf
