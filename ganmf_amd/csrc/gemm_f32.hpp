// fp32 MFMA GEMM for gfx950 (CDNA4): C[M,N] = op(A) . op(B) with fused epilogues.
//
// One kernel template serves every dense contraction of the GANMF step (SURVEY §8a rows a4-a6,
// a10, a11, a15).  The three operand-layout combinations the step needs are
//     NT  A [M,K] row-major, B [N,K] row-major      (generator / scoring GEMM U.V^T, dR.Wd^T, dE.We^T)
//     NN  A [M,K] row-major, B [K,N] row-major      (X.We, E.Wd, dF.V)
//     TN  A [K,M] row-major, B [K,N] row-major      (E^T.dR, inp^T.dE, dF^T.Ub)
// Design (MI355X_MICROARCH / cdna_hip_programming §3 "FP32-input MFMA", §5 "glds"):
//   * v_mfma_f32_32x32x2_f32 (exact fp32, 256 FLOP/clk/CU = the 157 TFLOP/s roof), 64-lane waves,
//     4 waves per workgroup in a 2x2 grid, each wave owning (BM/2)x(BN/2) of the block tile.
//   * operands go HBM/L2 -> LDS directly (global_load_lds_dwordx4, no VGPR staging) into an
//     NS-deep ring of K-tiles; the loop keeps NS-1 tiles in flight behind a COUNTED
//     s_waitcnt vmcnt and ONE raw s_barrier per K-tile (never __syncthreads(), which drains vmcnt).
//     The tail issues zero-page tiles so that the count stays a compile-time constant.
//   * the LDS image of a glds is lane-linear, so K-contiguous operands are stored unpadded
//     [row][BK] with the 16-byte chunk index XOR-swizzled by the row on the SOURCE address and
//     again on the read (conflict-free ds_read_b128; each read feeds 4 MFMAs).  The k index inside
//     an 8-wide chunk is permuted (lane half h, register r <-> k = 8c + 4h + r) identically for A
//     and B, which only re-orders the fp32 summation.  K-major operands stay [k][row] and are
//     read with conflict-free ds_read_b32 under the same k permutation.
//   * out-of-range rows / K-tail chunks are fetched from a 16-byte zero page (per-lane source
//     address), so no operand needs host-side padding beyond a leading dimension that is a
//     multiple of 64 floats with zeroed K-padding.
//   * 1-D grid with an XCD-aware, bijective block remap: the blocks of one XCD (blockIdx % 8) walk a
//     contiguous range of N panels with the M tiles innermost, so a B panel is fetched into that
//     XCD's L2 once.
//   * split-K writes fp32 partial slabs (no float atomics: bitwise reproducible, replicas stay
//     identical across GPUs); the caller reduces them with splitk_reduce_kernel.
#pragma once
#include <hip/hip_runtime.h>

namespace ganmf {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int LD_ALIGN = 64;  // floats; every leading dimension is a multiple of this

enum GemmEpi : int {
  EPI_STORE = 0,              // C = acc
  EPI_STORE_ONES_COL = 1,     // C = acc, and C[m, N] = 1 (the bias-folding ones column)
  EPI_SUB_AUX_SQ = 2,         // C = acc - aux[m,n];  per-block sum(C^2) -> sq_partials
  EPI_SUB_ROWSCALED_AUX = 3,  // C = acc - rowscale * aux[m,n]
};

struct GemmP {
  const float* A;
  const float* B;
  float* C;
  int lda, ldb, ldc;
  int M, N, K;
  const float* zero_page;  // >= 16 bytes of zeros in device memory
  int nsplit;              // >= 1
  int k_per_split;         // multiple of BK
  long long c_split_stride;
  int nbatch;              // >= 1; B is shared between batches
  long long a_batch_stride, c_batch_stride, aux_batch_stride;
  int epi;
  const float* aux;
  int ldaux;
  float rowscale_c;
  float* sq_partials;      // [nbatch][tiles_m * tiles_n]
  int tiles_m, tiles_n;
};

#define GANMF_WAIT_VMCNT(N) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory")

// One operand's share of a K-tile: R rows (M or N side) x BK k's, staged by 256 threads.
template <int R, int BK, bool KM>
struct Stage {
  static constexpr int F4 = R * BK / 4;          // float4 per tile
  static constexpr int NP = F4 / 256;            // glds per thread per tile
  static constexpr int SZ = R * BK;              // floats in LDS (unpadded)
  static constexpr int S = BK / 4;               // K-contig: 16-byte slots per row
  static constexpr int RC4 = R / 4;              // K-major: 16-byte slots per k-row
  static_assert(F4 % 256 == 0 && NP >= 1, "tile too small for 256 threads");
  static_assert(BK == 32 || BK == 64, "BK must be 32 or 64");

  // rows that share a 64-dword LDS bank row differ in bit 0 (BK=32) or not at all (BK=64)
  __device__ static inline int swz(int row) { return (row / (64 / BK)) & (S - 1); }

  const float* ptr[NP];   // per-lane source address of the next tile (KM: before validity select)
  int aux[NP];            // !KM: pointer increment per tile (0 for zero-page lanes); KM: k-row or -1

  __device__ inline void init(const float* __restrict__ base, int ld, int r0, int rlimit, int kbeg,
                              const float* __restrict__ zero, int tid) {
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      const int pos = j * 256 + tid;
      if constexpr (!KM) {
        const int row = pos / S, slot = pos % S;
        const int c4 = slot ^ swz(row);
        const bool ok = (r0 + row) < rlimit;
        ptr[j] = ok ? base + (size_t)(r0 + row) * ld + kbeg + 4 * c4 : zero;
        aux[j] = ok ? BK : 0;
      } else {
        const int krow = pos / RC4, c4 = pos % RC4;
        const bool ok = (r0 + 4 * c4) < ld;
        ptr[j] = base + (size_t)(kbeg + krow) * ld + r0 + 4 * c4;
        aux[j] = ok ? krow : -1;
      }
    }
  }

  // issue the glds of one tile into LDS at `s`; kleft = kend - k0 of this tile (<= 0: dummy tile)
  __device__ inline void issue(float* s, int ld, int kleft, const float* __restrict__ zero, int wave) {
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      const float* src;
      if constexpr (!KM) {
        src = kleft > 0 ? ptr[j] : zero;
        ptr[j] += aux[j];
      } else {
        src = (aux[j] >= 0 && aux[j] < kleft) ? ptr[j] : zero;
        ptr[j] += (size_t)BK * ld;
      }
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(s + (j * 256 + wave * 64) * 4),
                                       16, 0, 0);
    }
  }

  // fragment for the 32-row MFMA block starting at tile row `rb`, chunk c (8 k's), lane (i, h):
  // the 4 operands r = 0..3 with k = 8c + 4h + r
  __device__ static inline float4 frag(const float* __restrict__ s, int rb, int c, int i, int h) {
    if constexpr (!KM) {
      return *reinterpret_cast<const float4*>(s + (rb + i) * BK + 4 * ((2 * c + h) ^ swz(i)));
    } else {
      const float* q = s + (c * 8 + 4 * h) * R + rb + i;
      return make_float4(q[0], q[R], q[2 * R], q[3 * R]);
    }
  }
};

__device__ inline int xcd_remap(int bid, int nwg) {
  // blocks b and b+8 share an XCD (observed round-robin dispatch; speed only, never correctness)
  const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}

template <int BM, int BN, int BK, int NS, bool AKM, bool BKM>
__global__ __launch_bounds__(256) void gemm_f32_mfma(const GemmP p) {
  using SA = Stage<BM, BK, AKM>;
  using SB = Stage<BN, BK, BKM>;
  constexpr int WM = BM / 2, WN = BN / 2;
  constexpr int TM = WM / 32, TN = WN / 32;
  static_assert(TM >= 1 && TN >= 1, "wave tile must hold at least one 32x32 MFMA block");
  static_assert(NS >= 2 && NS <= 4, "ring depth");
  constexpr int BUF = SA::SZ + SB::SZ;            // ring slot b: A at smem + b*BUF, B right behind
  constexpr int LOADS = SA::NP + SB::NP;          // glds per wave per tile
  __shared__ __attribute__((aligned(16))) float smem[NS * BUF];   // the ONLY LDS object (cdna guide §5 item 4a)

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
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

  constexpr int NC = BK / 8;   // 8-wide k chunks per tile (even: chunk c uses fragment set c & 1)
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

  // prologue: tiles 0 .. NS-1 in flight (tiles past the K range come from the zero page so that
  // the vmcnt bookkeeping below stays a compile-time constant)
  int kleft = kend - kbeg;   // k's remaining from the next tile to issue
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    la.issue(smem + s * BUF, p.lda, kleft, p.zero_page, wave);
    lb.issue(smem + s * BUF + SA::SZ, p.ldb, kleft, p.zero_page, wave);
    kleft -= BK;
  }
  GANMF_WAIT_VMCNT((NS - 1) * LOADS);   // tile 0 of this wave has landed ...
  __builtin_amdgcn_s_barrier();         // ... and of every other wave
  load_frags(0, smem, 0);

  int slot = 0;   // ring slot of tile `it`
  for (int it = 0; it < nt; ++it) {
    const float* __restrict__ cur = smem + slot * BUF;
    const int nslot = (slot + 1 == NS) ? 0 : slot + 1;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      if (c + 1 < NC) {
        load_frags((c + 1) & 1, cur, c + 1);     // next chunk's fragments under this chunk's MFMAs
        __builtin_amdgcn_sched_barrier(0);       // keep the reads ahead of the MFMAs (hipcc sinks them)
      } else {
        // Last chunk: its fragments are in registers once lgkmcnt drains, so this wave no longer
        // reads slot `slot`.  Tile it+1 has landed when at most NS-2 younger tiles are outstanding;
        // after the barrier that holds for every wave and slot `slot` is free for tile it+NS.
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        GANMF_WAIT_VMCNT((NS - 2) * LOADS);
        __builtin_amdgcn_s_barrier();
        la.issue(smem + slot * BUF, p.lda, kleft, p.zero_page, wave);
        lb.issue(smem + slot * BUF + SA::SZ, p.ldb, kleft, p.zero_page, wave);
        kleft -= BK;
        load_frags(0, smem + nslot * BUF, 0);    // first fragments of tile it+1 under the last MFMAs
        __builtin_amdgcn_sched_barrier(0);
      }
      mfmas(c & 1);
    }
    slot = nslot;
  }
  // drain the (zero-page) tiles still in flight before LDS is reused / the block exits
  GANMF_WAIT_VMCNT(0);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  // ---- epilogue: C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
  float* __restrict__ C = p.C + (size_t)sp * p.c_split_stride + (size_t)bz * p.c_batch_stride;
  const float* __restrict__ aux = p.aux ? p.aux + (size_t)bz * p.aux_batch_stride : nullptr;
  const int epi = p.epi;
  float sq = 0.f;
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b) {
      const int col = n0 + wc * WN + b * 32 + li;
      const bool colok = col < p.N;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wr * WM + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (row < p.M) {
          if (colok) {
            float v = acc[a][b][r];
            if (epi == EPI_SUB_AUX_SQ) {
              v -= aux[(size_t)row * p.ldaux + col];
              sq += v * v;
            } else if (epi == EPI_SUB_ROWSCALED_AUX) {
              v -= p.rowscale_c * aux[(size_t)row * p.ldaux + col];
            }
            C[(size_t)row * p.ldc + col] = v;
          } else if (epi == EPI_STORE_ONES_COL && col == p.N) {
            C[(size_t)row * p.ldc + col] = 1.0f;
          }
        }
      }
    }
  if (epi == EPI_SUB_AUX_SQ) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
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
  RED_ONES_COL = 1,  // plain, and out[m, N] = 1 (bias-folding ones column of the encodings)
  RED_ROWSCALE = 2,  // * rowscale[m]
  RED_G_DE = 3,      // rowscale_c*sum + cfm*(Ef - Er)[m,n]; per-block sum((Ef-Er)^2) -> sq_partials
};

struct RedP {
  const float* part;
  long long split_stride;
  int nsplit;
  float* out;
  int ld;       // shared by part / out / er / ef
  int M, N;
  int epi;
  const float* rowscale;  // [M] or nullptr -> rowscale_c
  float rowscale_c;
  const float* er;
  const float* ef;
  float cfm;              // alpha*2/(B_global*e)
  float* sq_partials;     // [gridDim.x]
};

__global__ __launch_bounds__(256) void splitk_reduce_kernel(const RedP p) {
  const int n4 = p.N / 4 + 1;   // covers column N (the ones column) as well; ld >= N + 1 rounded to 64
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
    if (p.epi == RED_ROWSCALE) {
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
      if (c + j >= p.N) o[j] = (p.epi == RED_ONES_COL && c + j == p.N) ? 1.f : 0.f;  // K-padding stays zero
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
// Tile configurations: {64x64, BK 64} for skinny problems, {128x128, BK 32} otherwise; both have
// 32 KiB ring slots.  NS = 3 slots (96 KiB, one workgroup per CU, two K-tiles in flight).
constexpr int GEMM_K_ALIGN = 64;   // split-K slices are multiples of this (>= every BK)
constexpr int GEMM_NS = 3;

inline int gemm_pick_tile(int M, int N, int nsplit, int nbatch) {
  const long long t128 = (long long)((M + 127) / 128) * ((N + 127) / 128) * nsplit * nbatch;
  return t128 >= 192 ? 128 : 64;
}

template <int BM, int BN, int BK, int NS>
inline hipError_t gemm_launch_t(hipStream_t st, GemmP& p, bool akm, bool bkm) {
  p.tiles_m = (p.M + BM - 1) / BM;
  p.tiles_n = (p.N + BN - 1) / BN;
  const int grid = p.tiles_m * p.tiles_n * p.nsplit * p.nbatch;
  if (grid <= 0) return hipSuccess;
  if (!akm && !bkm) hipLaunchKernelGGL((gemm_f32_mfma<BM, BN, BK, NS, false, false>), dim3(grid), dim3(256), 0, st, p);
  else if (!akm && bkm) hipLaunchKernelGGL((gemm_f32_mfma<BM, BN, BK, NS, false, true>), dim3(grid), dim3(256), 0, st, p);
  else if (akm && bkm) hipLaunchKernelGGL((gemm_f32_mfma<BM, BN, BK, NS, true, true>), dim3(grid), dim3(256), 0, st, p);
  else return hipErrorInvalidValue;  // TT is not needed by the GANMF step
  return hipGetLastError();
}

inline hipError_t gemm_launch(hipStream_t st, GemmP& p, bool akm, bool bkm, int tile) {
  if (p.nsplit < 1) p.nsplit = 1;
  if (p.nbatch < 1) p.nbatch = 1;
  if (p.nsplit == 1) p.k_per_split = ((p.K + GEMM_K_ALIGN - 1) / GEMM_K_ALIGN) * GEMM_K_ALIGN;
  if (!p.zero_page || (p.lda % LD_ALIGN) || (p.ldb % LD_ALIGN)) return hipErrorInvalidValue;
  if (tile == 0) tile = gemm_pick_tile(p.M, p.N, p.nsplit, p.nbatch);
  if (tile == 128) return gemm_launch_t<128, 128, 32, GEMM_NS>(st, p, akm, bkm);
  return gemm_launch_t<64, 64, 64, GEMM_NS>(st, p, akm, bkm);
}

// number of tiles the partial buffer must hold for EPI_SUB_AUX_SQ (worst case tile = 64)
inline int gemm_max_tiles(int M, int N) { return ((M + 63) / 64) * ((N + 63) / 64); }

inline void split_plan(int K, int want, int& nsplit, int& kps) {
  int chunks = (K + GEMM_K_ALIGN - 1) / GEMM_K_ALIGN;
  if (want < 1) want = 1;
  if (want > chunks) want = chunks;
  int cps = (chunks + want - 1) / want;
  kps = cps * GEMM_K_ALIGN;
  nsplit = (chunks + cps - 1) / cps;
}

}  // namespace ganmf
