// fp32 MFMA GEMM for gfx950 (CDNA4): C[M,N] = op(A) . op(B) with fused epilogues.
//
// One kernel template serves every dense contraction of the GANMF step (SURVEY §8a rows a4-a6,
// a10, a11, a15): the three operand-layout combinations the step needs are
//     NT  A [M,K] row-major, B [N,K] row-major      (generator / scoring GEMM U.V^T, dR.Wd^T, dE.We^T)
//     NN  A [M,K] row-major, B [K,N] row-major      (X.We, E.Wd, dF.V)
//     TN  A [K,M] row-major, B [K,N] row-major      (E^T.dR, inp^T.dE, dF^T.Ub)
// Design (MI355X_MICROARCH / cdna_hip_programming §3 "FP32-input MFMA"):
//   * v_mfma_f32_32x32x2_f32 (exact fp32, 256 FLOP/clk/CU = the 157 TFLOP/s roof), 64-lane waves,
//     4 waves per workgroup in a 2x2 grid, each wave owning (BM/2)x(BN/2) of the block tile.
//   * operands staged global -> registers -> LDS, double buffered, one barrier per K-tile; global
//     loads of tile t+1 are issued before the MFMAs of tile t.
//   * K-contiguous operands are kept K-contiguous in LDS (row stride BK+4 floats: conflict-free
//     ds_read_b128); each 16-byte read feeds 4 MFMAs.  The k index inside an 8-wide chunk is
//     permuted (lane half h, register r  <->  k = 8c + 4h + r) identically for A and B, which only
//     re-orders the fp32 summation.  K-major operands stay K-major in LDS and are read with
//     conflict-free ds_read_b32 using the same k permutation.
//   * 1-D grid with an XCD-aware, bijective block remap: the blocks of one XCD (blockIdx % 8) walk a
//     contiguous range of N panels with the M tiles innermost, so a B panel is fetched into that
//     XCD's L2 once.
//   * split-K writes fp32 partial slabs (no float atomics: bitwise reproducible, replicas stay
//     identical across GPUs); the caller reduces them with splitk_reduce_kernel.
#pragma once
#include <hip/hip_runtime.h>

namespace ganmf {

typedef float f32x16 __attribute__((ext_vector_type(16)));

enum GemmEpi : int {
  EPI_STORE = 0,              // C = acc
  EPI_BIAS = 1,               // C = acc + bias[n]
  EPI_BIAS_SUB_AUX_SQ = 2,    // C = acc + bias[n] - aux[m,n];  per-block sum(C^2) -> sq_partials
  EPI_SUB_ROWSCALED_AUX = 3,  // C = acc - rowscale[m] * aux[m,n]
};

struct GemmP {
  const float* A;
  const float* B;
  float* C;
  int lda, ldb, ldc;
  int M, N, K;
  const float* kscale;  // optional, K-major A only: A(k, :) is multiplied by kscale[k] while staged
  int nsplit;           // >= 1
  int k_per_split;      // multiple of BK
  long long c_split_stride;
  int nbatch;           // >= 1; B is shared between batches
  long long a_batch_stride, c_batch_stride, aux_batch_stride;
  int epi;
  const float* bias;
  const float* aux;
  int ldaux;
  const float* rowscale;   // [M] per batch, or nullptr -> rowscale_c
  float rowscale_c;
  int rowscale_batch_stride;
  float* sq_partials;   // [nbatch][tiles_m * tiles_n]
  int tiles_m, tiles_n;
};

template <int R, int BK, bool KM>
struct TileLoader {
  static constexpr int C4 = KM ? R / 4 : BK / 4;   // float4 per staged row
  static constexpr int ROWS = KM ? BK : R;
  static constexpr int RPP = 256 / C4;             // rows per pass of the 256 threads
  static constexpr int NP = ROWS / RPP;
  static constexpr int LD = KM ? R + 4 : BK + 4;   // LDS row stride in floats
  static constexpr int SZ = ROWS * LD;
  static_assert(256 % C4 == 0 && ROWS % RPP == 0 && NP >= 1, "tile/loader mismatch");

  __device__ static inline void load(float4 (&reg)[NP], const float* __restrict__ base, int ld, int r0,
                                     int rlimit, int k0, int kend, const float* __restrict__ kscale, int tid) {
    const int tr = tid / C4, tc = tid % C4;
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if constexpr (!KM) {
        const int row = r0 + tr + j * RPP;
        const int k = k0 + tc * 4;
        if (row < rlimit && k < kend) {
          v = *reinterpret_cast<const float4*>(base + (size_t)row * ld + k);
          if (k + 3 >= kend) {  // ragged K tail
            if (k + 1 >= kend) v.y = 0.f;
            if (k + 2 >= kend) v.z = 0.f;
            v.w = 0.f;
          }
        }
      } else {
        const int k = k0 + tr + j * RPP;
        const int col = r0 + tc * 4;
        if (k < kend && col < ld) {
          v = *reinterpret_cast<const float4*>(base + (size_t)k * ld + col);
          if (kscale) {
            const float s = kscale[k];
            v.x *= s; v.y *= s; v.z *= s; v.w *= s;
          }
        }
      }
      reg[j] = v;
    }
  }

  __device__ static inline void store(const float4 (&reg)[NP], float* __restrict__ s, int tid) {
    const int tr = tid / C4, tc = tid % C4;
#pragma unroll
    for (int j = 0; j < NP; ++j)
      *reinterpret_cast<float4*>(s + (tr + j * RPP) * LD + tc * 4) = reg[j];
  }

  // fragment for MFMA block starting at tile row `rb`, chunk c (8 k's), lane (i, h):
  // returns the 4 operands r=0..3 with k = 8c + 4h + r
  __device__ static inline float4 frag(const float* __restrict__ s, int rb, int c, int i, int h) {
    if constexpr (!KM) {
      return *reinterpret_cast<const float4*>(s + (rb + i) * LD + c * 8 + 4 * h);
    } else {
      const float* q = s + (c * 8 + 4 * h) * LD + rb + i;
      return make_float4(q[0], q[LD], q[2 * LD], q[3 * LD]);
    }
  }
};

__device__ inline int xcd_remap(int bid, int nwg) {
  // blocks b and b+8 share an XCD (observed round-robin dispatch; speed only, never correctness)
  const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}

template <int BM, int BN, int BK, bool AKM, bool BKM>
__global__ __launch_bounds__(256) void gemm_f32_mfma(const GemmP p) {
  using LA = TileLoader<BM, BK, AKM>;
  using LB = TileLoader<BN, BK, BKM>;
  constexpr int WM = BM / 2, WN = BN / 2;
  constexpr int TM = WM / 32, TN = WN / 32;
  static_assert(TM >= 1 && TN >= 1, "wave tile must hold at least one 32x32 MFMA block");
  __shared__ __attribute__((aligned(16))) float smem[2 * (LA::SZ + LB::SZ)];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int li = lane & 31, lh = lane >> 5;

  int t = xcd_remap(blockIdx.x, gridDim.x);
  const int tm = t % p.tiles_m; t /= p.tiles_m;
  const int tn = t % p.tiles_n; t /= p.tiles_n;
  const int sp = t % p.nsplit;
  const int bz = t / p.nsplit;

  const int m0 = tm * BM, n0 = tn * BN;
  const int kbeg = sp * p.k_per_split;
  const int kend = min(p.K, kbeg + p.k_per_split);
  const int nt = (kend - kbeg + BK - 1) / BK;

  const float* __restrict__ A = p.A + (size_t)bz * p.a_batch_stride;
  const float* __restrict__ B = p.B;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  float4 ra[LA::NP], rb[LB::NP];
  constexpr int BUF = LA::SZ + LB::SZ;  // buffer b: A at smem + b*BUF, B at smem + b*BUF + LA::SZ

  if (nt > 0) {
    LA::load(ra, A, p.lda, m0, p.M, kbeg, kend, p.kscale, tid);
    LB::load(rb, B, p.ldb, n0, p.N, kbeg, kend, nullptr, tid);
    LA::store(ra, smem, tid);
    LB::store(rb, smem + LA::SZ, tid);
  }
  __syncthreads();

  for (int it = 0; it < nt; ++it) {
    const int cur = it & 1;
    if (it + 1 < nt) {
      const int k0 = kbeg + (it + 1) * BK;
      LA::load(ra, A, p.lda, m0, p.M, k0, kend, p.kscale, tid);
      LB::load(rb, B, p.ldb, n0, p.N, k0, kend, nullptr, tid);
    }
    const float* __restrict__ a_s = smem + cur * BUF;
    const float* __restrict__ b_s = smem + cur * BUF + LA::SZ;
#pragma unroll
    for (int c = 0; c < BK / 8; ++c) {
      float4 fa[TM], fb[TN];
#pragma unroll
      for (int a = 0; a < TM; ++a) fa[a] = LA::frag(a_s, wr * WM + a * 32, c, li, lh);
#pragma unroll
      for (int b = 0; b < TN; ++b) fb[b] = LB::frag(b_s, wc * WN + b * 32, c, li, lh);
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) {
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a].x, fb[b].x, acc[a][b], 0, 0, 0);
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a].y, fb[b].y, acc[a][b], 0, 0, 0);
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a].z, fb[b].z, acc[a][b], 0, 0, 0);
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a].w, fb[b].w, acc[a][b], 0, 0, 0);
        }
    }
    if (it + 1 < nt) {
      LA::store(ra, smem + (cur ^ 1) * BUF, tid);
      LB::store(rb, smem + (cur ^ 1) * BUF + LA::SZ, tid);
    }
    __syncthreads();
  }

  // ---- epilogue: C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
  float* __restrict__ C = p.C + (size_t)sp * p.c_split_stride + (size_t)bz * p.c_batch_stride;
  const float* __restrict__ aux = p.aux ? p.aux + (size_t)bz * p.aux_batch_stride : nullptr;
  const float* __restrict__ rowscale = p.rowscale ? p.rowscale + bz * p.rowscale_batch_stride : nullptr;
  const int epi = p.epi;
  float sq = 0.f;
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b) {
      const int col = n0 + wc * WN + b * 32 + li;
      const bool colok = col < p.N;
      float bias = 0.f;
      if ((epi == EPI_BIAS || epi == EPI_BIAS_SUB_AUX_SQ) && colok) bias = p.bias[col];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wr * WM + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (row < p.M && colok) {
          float v = acc[a][b][r];
          if (epi == EPI_BIAS) {
            v += bias;
          } else if (epi == EPI_BIAS_SUB_AUX_SQ) {
            v = (v + bias) - aux[(size_t)row * p.ldaux + col];
            sq += v * v;
          } else if (epi == EPI_SUB_ROWSCALED_AUX) {
            v -= (rowscale ? rowscale[row] : p.rowscale_c) * aux[(size_t)row * p.ldaux + col];
          }
          C[(size_t)row * p.ldc + col] = v;
        }
      }
    }
  if (epi == EPI_BIAS_SUB_AUX_SQ) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
    // smem is free again: every wave passed the K-loop's final barrier
    if (lane == 0) smem[wave] = sq;
    __syncthreads();
    if (tid == 0)
      p.sq_partials[(size_t)bz * p.tiles_m * p.tiles_n + tn * p.tiles_m + tm] =
          (smem[0] + smem[1]) + (smem[2] + smem[3]);
  }
}

// Reduce split-K slabs.  out[m,n] = sum_s part[s][m,n], then one of:
enum RedEpi : int {
  RED_PLAIN = 0,
  RED_BIAS = 1,      // + bias[n]
  RED_ROWSCALE = 2,  // * rowscale[m]
  RED_G_DE = 3,      // rowscale[m]*sum + cfm*(Ef - Er)[m,n]; per-block sum((Ef-Er)^2) -> sq_partials
};

struct RedP {
  const float* part;
  long long split_stride;
  int nsplit;
  float* out;
  int ld;       // shared by part / out / er / ef
  int M, N;
  int epi;
  const float* bias;
  const float* rowscale;  // [M] or nullptr -> rowscale_c
  float rowscale_c;
  const float* er;
  const float* ef;
  float cfm;              // alpha*2/(B_global*e)
  float* sq_partials;     // [gridDim.x]
};

__global__ __launch_bounds__(256) void splitk_reduce_kernel(const RedP p) {
  const int n4 = (p.N + 3) >> 2;  // ld % 4 == 0 and pad columns are never consumed as K data unmasked
  const long long total = (long long)p.M * n4;
  float sq = 0.f;
  const float cfm = p.cfm;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int m = (int)(idx / n4), c = (int)(idx % n4) * 4;
    const size_t off = (size_t)m * p.ld + c;
    float4 s = *reinterpret_cast<const float4*>(p.part + off);
    for (int k = 1; k < p.nsplit; ++k) {
      const float4 q = *reinterpret_cast<const float4*>(p.part + (size_t)k * p.split_stride + off);
      s.x += q.x; s.y += q.y; s.z += q.z; s.w += q.w;
    }
    float o[4] = {s.x, s.y, s.z, s.w};
    if (p.epi == RED_BIAS) {
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] += (c + j < p.N) ? p.bias[c + j] : 0.f;
    } else if (p.epi == RED_ROWSCALE) {
      const float r = p.rowscale ? p.rowscale[m] : p.rowscale_c;
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] *= r;
    } else if (p.epi == RED_G_DE) {
      const float r = p.rowscale ? p.rowscale[m] : p.rowscale_c;
      const float4 er = *reinterpret_cast<const float4*>(p.er + off);
      const float4 ef = *reinterpret_cast<const float4*>(p.ef + off);
      const float d[4] = {ef.x - er.x, ef.y - er.y, ef.z - er.z, ef.w - er.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        o[j] = r * o[j] + cfm * d[j];
        if (c + j < p.N) sq += d[j] * d[j];
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (c + j >= p.N) o[j] = 0.f;  // keep K-padding columns exactly zero
    *reinterpret_cast<float4*>(p.out + off) = make_float4(o[0], o[1], o[2], o[3]);
  }
  if (p.epi == RED_G_DE) {
    __shared__ float red[4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sq;
    __syncthreads();
    if (threadIdx.x == 0) p.sq_partials[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
  }
}

// ---- host-side launcher ---------------------------------------------------------------------
struct GemmLaunch {
  int tile;     // 64 or 128
  int tiles_m, tiles_n;
  int grid;
};

inline int gemm_pick_tile(int M, int N, int nsplit, int nbatch) {
  const long long t128 = (long long)((M + 127) / 128) * ((N + 127) / 128) * nsplit * nbatch;
  return t128 >= 192 ? 128 : 64;
}

template <int BM, int BN, int BK>
inline hipError_t gemm_launch_t(hipStream_t st, GemmP& p, bool akm, bool bkm) {
  p.tiles_m = (p.M + BM - 1) / BM;
  p.tiles_n = (p.N + BN - 1) / BN;
  const int grid = p.tiles_m * p.tiles_n * p.nsplit * p.nbatch;
  if (grid <= 0) return hipSuccess;
  if (!akm && !bkm) hipLaunchKernelGGL((gemm_f32_mfma<BM, BN, BK, false, false>), dim3(grid), dim3(256), 0, st, p);
  else if (!akm && bkm) hipLaunchKernelGGL((gemm_f32_mfma<BM, BN, BK, false, true>), dim3(grid), dim3(256), 0, st, p);
  else if (akm && bkm) hipLaunchKernelGGL((gemm_f32_mfma<BM, BN, BK, true, true>), dim3(grid), dim3(256), 0, st, p);
  else return hipErrorInvalidValue;  // TT is not needed by the GANMF step
  return hipGetLastError();
}

constexpr int GEMM_BK = 32;

inline hipError_t gemm_launch(hipStream_t st, GemmP& p, bool akm, bool bkm, int tile) {
  if (p.nsplit < 1) p.nsplit = 1;
  if (p.nbatch < 1) p.nbatch = 1;
  if (p.nsplit == 1) p.k_per_split = ((p.K + GEMM_BK - 1) / GEMM_BK) * GEMM_BK;
  if (tile == 0) tile = gemm_pick_tile(p.M, p.N, p.nsplit, p.nbatch);
  if (tile == 128) return gemm_launch_t<128, 128, GEMM_BK>(st, p, akm, bkm);
  return gemm_launch_t<64, 64, GEMM_BK>(st, p, akm, bkm);
}

// number of tiles the partial buffer must hold for EPI_BIAS_SUB_AUX_SQ (worst case tile = 64)
inline int gemm_max_tiles(int M, int N) { return ((M + 63) / 64) * ((N + 63) / 64); }

inline void split_plan(int K, int want, int& nsplit, int& kps) {
  int chunks = (K + GEMM_BK - 1) / GEMM_BK;
  if (want < 1) want = 1;
  if (want > chunks) want = chunks;
  int cps = (chunks + want - 1) / want;
  kps = cps * GEMM_BK;
  nsplit = (chunks + cps - 1) / cps;
}

}  // namespace ganmf
