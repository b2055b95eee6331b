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
//     Steady-state tiles and the last NS-1 tiles are two loops, each with a compile-time wait count; nothing is
//     loaded past the K range.
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
// The same contractions on the bf16 matrix cores (fp32-accurate operand split, or mixed precision): gemm_bf16s.hpp,
// chosen per GEMM by gemm_plan below; both kernels share gemm_epilogue.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#include <type_traits>

namespace ganmf {

// Every kernel of the library is launched through GANMF_LAUNCH.  While a profiling scope is armed (ganmf_profile_enable:
// bench.py's per-class table and its `roofline` object) the FIRST launch inside the scope goes through hipExtLaunchKernelGGL
// with the scope's two events, which the runtime then stamps with the kernel's own start and end -- the same interval
// rocprofv3 reports -- instead of bracketing the launch with event records (which adds the ~2 us the two marker packets
// and the launch gap take).  Outside profiling it is a plain hipLaunchKernelGGL.
struct LaunchProf { hipEvent_t start = nullptr, stop = nullptr; int count = 0; };
inline LaunchProf& launch_prof() { static thread_local LaunchProf lp; return lp; }
// A stream fork without a marker packet: the NEXT launch carries this event as its completion ("stop") event, so another stream
// can wait for exactly that kernel; recording an event behind the kernel instead puts a marker packet into the producing
// stream, which showed as ~5 us of idle lane per fork in the data-parallel step's timeline (profiles/r03_dp_timeline.md).
inline hipEvent_t& launch_stop_event() { static thread_local hipEvent_t ev = nullptr; return ev; }
inline long long& launch_count() { static thread_local long long n = 0; return n; }      // launches of this host thread (fork_arm / fork_wait)

template <class F, class... Args>
inline void launch_kernel(F kernel, const dim3& grid, const dim3& block, unsigned shmem, hipStream_t st, Args... args) {
  LaunchProf& lp = launch_prof();
  hipEvent_t& stop = launch_stop_event();
  ++launch_count();
  if (lp.start && lp.count++ == 0) hipExtLaunchKernelGGL(kernel, grid, block, shmem, st, lp.start, lp.stop, 0, args...);
  else if (stop) { hipExtLaunchKernelGGL(kernel, grid, block, shmem, st, nullptr, stop, 0, args...); stop = nullptr; }
  else hipLaunchKernelGGL(kernel, grid, block, shmem, st, args...);
}
#define GANMF_LAUNCH(...) ::ganmf::launch_kernel(__VA_ARGS__)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// MFMA arithmetic of the K loop (GemmPlan::mode):
//   MFMA_F32    gemm_f32_mfma below: v_mfma_f32_32x32x2_f32 on the fp32 operands (256 FLOP/clk/CU)
//   MFMA_BF16X3 gemm_bf16s.hpp: every fp32 operand is split EXACTLY into three bf16 pieces x = hi + mid + lo
//               (round-to-nearest at each level, see split_bf16x3) and a.b is accumulated in fp32 from the six
//               piece products of weight >= 2^-18 (hi.hi, hi.mid, mid.hi, hi.lo, lo.hi, mid.mid) on
//               v_mfma_f32_32x32x16_bf16 (4096 FLOP/clk/CU): 6/16 of the fp32 MFMA time; the three dropped
//               products sum to <= 2^-26 |a||b|, below one fp32 rounding of the product, and are unbiased.
//   MFMA_BF16   gemm_bf16s.hpp: operands rounded to one bf16 (round to nearest even), fp32 accumulate (master weights and
//               Adam stay fp32).
//   MFMA_F16    gemm_bf16s.hpp: operands rounded to one IEEE fp16 (v_cvt_pk_f16_f32, RNE) for v_mfma_f32_32x32x16_f16,
//               fp32 accumulate: the mixed-precision variant BASELINE configs[4] names.  fp16 has 11 significant bits
//               (bf16: 8) but a 5-bit exponent: operands that carry the 1/(B.N) of the loss gradients are multiplied by a
//               power of two when they are converted (GemmP::a_scale / b_scale, exact) and the accumulator is scaled back
//               in fp32 before the epilogue -- static loss scaling per GEMM; converted values are clamped to +-65504.
enum MfmaMode : int { MFMA_AUTO = -1, MFMA_F32 = 0, MFMA_BF16 = 1, MFMA_F16 = 2, MFMA_BF16X3 = 3 };

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

// two fp32 -> two bf16 in one dword (x0 in the low half), round to nearest even: v_cvt_pk_bf16_f32
// (inline asm: through __builtin_convertvector hipcc first builds a float2, and builds it through scratch / LDS
// when x0 and x1 live in registers that are not adjacent)
__device__ inline unsigned cvt_pk_bf16(float x0, float x1) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(x0), "v"(x1));
  return r;
}

// the same conversion carrying a scalar token as a fourth operand (never read by the instruction: it is text in a comment): the
// conversion cannot be scheduled in front of whatever produced the token -- how the 16-wave kernel orders the FIRST reads of its prefetch
// registers behind its hand-counted vmcnt wait (gemm_bf16k.hpp bf16k_wait_vm) without a select per element
__device__ inline unsigned cvt_pk_bf16_tok(float x0, float x1, int tok) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2 ; behind wait token %3" : "=v"(r) : "v"(x0), "v"(x1), "s"(tok));
  return r;
}
__device__ inline void split_bf16x3_tok(float x0, float x1, unsigned& hi, unsigned& mid, unsigned& lo, int tok) {
  hi = cvt_pk_bf16_tok(x0, x1, tok);
  float r0 = x0 - __uint_as_float(hi << 16), r1 = x1 - __uint_as_float(hi & 0xffff0000u);
  mid = cvt_pk_bf16(r0, r1);
  r0 -= __uint_as_float(mid << 16);
  r1 -= __uint_as_float(mid & 0xffff0000u);
  lo = cvt_pk_bf16(r0, r1);
}

// two fp32 -> two IEEE fp16 in one dword (x0 in the low half), round to nearest even, clamped to the finite fp16 range
// (an overflow would otherwise become an infinity and poison the accumulators)
__device__ inline unsigned cvt_pk_f16(float x0, float x1) {
  unsigned r;
  x0 = __builtin_amdgcn_fmed3f(x0, -65504.f, 65504.f);
  x1 = __builtin_amdgcn_fmed3f(x1, -65504.f, 65504.f);
  asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(x0), "v"(x1));
  return r;
}

// two fp32 -> three dwords of two bf16 each with x = hi + mid + lo EXACTLY:
// hi = rne(x) keeps 8 bits, r1 = x - hi is exact (<= 16 bits), mid = rne(r1), r2 = r1 - mid is exact and has <= 8
// significant bits, so lo = r2.  Round-to-nearest keeps |mid| <= 2^-9 |x|, |lo| <= 2^-18 |x| with signs that are
// not tied to the sign of x: the three dropped piece products are <= 2^-26 |a||b| together and unbiased
// (truncation would leave residues of the sign of x and a systematic 2^-23 |a||b| under-estimate that long
// sums with cancellation amplify).  Scalar operands on purpose: the two elements of a pair come from different
// registers for K-major operands, and hipcc builds a cross-register float2 through scratch / LDS.
__device__ inline void split_bf16x3(float x0, float x1, unsigned& hi, unsigned& mid, unsigned& lo) {
  hi = cvt_pk_bf16(x0, x1);
  float r0 = x0 - __uint_as_float(hi << 16), r1 = x1 - __uint_as_float(hi & 0xffff0000u);
  mid = cvt_pk_bf16(r0, r1);
  r0 -= __uint_as_float(mid << 16);
  r1 -= __uint_as_float(mid & 0xffff0000u);
  lo = cvt_pk_bf16(r0, r1);
}

template <int I, int N, class F>
__device__ inline void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

constexpr int LD_ALIGN = 64;  // floats; every leading dimension is a multiple of this

// Epilogue applied to the K-complete sum; shared by the GEMM kernel (nsplit == 1) and the
// split-K reduce kernel (nsplit > 1), so every GEMM of the step can be split along K freely.
enum GemmEpi : int {
  EPI_STORE = 0,           // C = acc
  EPI_SUB_AUX_SQ = 1,      // C = acc - aux[m,n];            sum(C^2) -> sq_partials
  EPI_SUB_SCALED_AUX = 2,  // C = acc - c * aux[m,n]
  EPI_ROWSCALE = 3,        // C = rowscale[m] * acc
  EPI_G_DE = 4,            // C = c * acc + cfm * (ef - er)[m,n];   sum((ef-er)^2) -> sq_partials
  EPI_ACT = 5,             // C = act(acc)                      (DisGANMF hidden layers)
  EPI_MUL_ACTGRAD = 6,     // C = acc * act'(aux[m,n]) from the layer OUTPUT aux   (DisGANMF backward)
  EPI_ADAM = 7,            // acc is the gradient: TF ApplyAdam on theta/m/v in place, nothing stored to C;
                           // sum(theta_old^2) -> sq_partials.  Never split along K.
};

constexpr float ADAM_B1 = 0.9f, ADAM_B2 = 0.999f, ADAM_EPS = 1e-8f;

enum ActKind : int { ACT_LINEAR = 0, ACT_TANH = 1, ACT_RELU = 2, ACT_SIGMOID = 3 };

__device__ inline float act_apply(int act, float z) {
  switch (act) {
    case ACT_TANH: return tanhf(z);
    case ACT_RELU: return fmaxf(z, 0.f);
    case ACT_SIGMOID: return 1.f / (1.f + expf(-z));
    default: return z;
  }
}
__device__ inline float act_grad_out(int act, float a) {
  switch (act) {
    case ACT_TANH: return 1.f - a * a;
    case ACT_RELU: return a > 0.f ? 1.f : 0.f;
    case ACT_SIGMOID: return a * (1.f - a);
    default: return 1.f;
  }
}

struct EpiD {
  int kind;
  const float* aux;        // [.., ldaux] (per batch: + bz * aux_batch_stride)
  int ldaux;
  long long aux_batch_stride;
  const float* rowscale;   // [M]
  float c;
  const float* er;         // [M, ldc]
  const float* ef;
  float cfm;
  float* sq_partials;      // [nbatch][sq_stride]
  int sq_stride;
  int sq_m_half;           // > 0 (unsplit, unbatched products only): the rows are TWO stacked batches of sq_m_half row tiles each
                           // ([real ; generated] of the decode GEMM); partials are filed as the two-batch form files them --
                           // batch = tm / sq_m_half, slot tn * sq_m_half + tm % sq_m_half -- so the consumer sums the same numbers
                           // in the same order, while the weight panel of a tile column is fetched once for both batches
  int act;                 // ActKind for EPI_ACT / EPI_MUL_ACTGRAD
  const float* r1_u;       // EPI_ACT: optional fp32 rank-1 term  acc += r1_u[m * r1_ld] * r1_w[n]  before the activation
  const float* r1_w;       //          (DisGANMF's float(uid) input column kept OUT of a low-precision K loop)
  int r1_ld;
  float* adam_theta;       // EPI_ADAM: parameter and moments, same [M, ldc] geometry as C
  float* adam_theta_out;   //   where the updated parameter goes (nullptr: in place).  A second buffer lets another GEMM of
                           //   the SAME launch still read the old parameter (gemm_multi.hpp: gUb reads V while gV updates it)
  float* adam_m;
  float* adam_v;
  const float* adam_alpha; // device scalar lr_t of the step in flight
  float adam_reg;          // gradient += adam_reg * theta  (the L2 term of the loss)
  // ---- sparse-aware real path of the discriminator step (GANMF.py:183-187): the real rows X stay CSR
  // EPI_SUB_AUX_SQ, batch 0 (the real path): the subtracted operand X[m, n] is looked up in the CSR rows of the batch
  // (row m of the tile = CSR row csr_rows[m]; columns sorted inside a row) instead of a dense aux matrix -- subtracting 0.0f
  // is exact, so the residual and its sum of squares equal the dense path's bit for bit.
  const long long* csr_indptr;
  const int* csr_indices;
  const float* csr_data;
  const int* csr_rows;
  // EPI_ADAM / EPI_STORE of the encoder gradient gWe_ext = [X;F|1]^T . dE: the GEMM runs over the generated rows only and the real
  // rows' contribution  S[j, :] = sum_b X[b, j] * dE_r[b, :]  -- formed beforehand by csc_rows_kernel (kernels.hpp) from the CSC form
  // of the matrix, in a fixed order -- is added here, before Adam / the store.  Rows sp_bias_row .. sp_bias_row + sp_bias_parts - 1
  // of S hold partial column sums of dE_r (the real rows' share of the encoder-bias gradient): added in index order to row
  // sp_bias_row of the gradient.
  const float* sp_rows;    // S [sp_bias_row + sp_bias_parts, sp_ld]; nullptr: dense path
  int sp_ld, sp_bias_row, sp_bias_parts;
  // the slab-sum kernel only (splitk_reduce_body): the finished values are ALSO filed as their three bf16 pieces, row-major with C's
  // geometry, piece q at planes + q * plane_stride -- the operand form of wgrad_stream.hpp (nullptr: no planes)
  unsigned short* planes;
  long long plane_stride;
};

// X[batch row m, column col .. col + 3] of the CSR rows of the batch (EpiD::csr_*): lower bound on the sorted column indices
__device__ inline float4 csr_quad(const EpiD& e, int m, int col) {
  const int r = e.csr_rows[m];
  long long lo = e.csr_indptr[r];
  const long long en = e.csr_indptr[r + 1];
  long long hi = en;
  while (lo < hi) {
    const long long mid = (lo + hi) >> 1;
    if (e.csr_indices[mid] < col) lo = mid + 1; else hi = mid;
  }
  float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
  for (long long j = lo; j < en; ++j) {
    const int c = e.csr_indices[j] - col;
    if (c > 3) break;
    const float v = e.csr_data[j];
    if (c == 0) x.x = v; else if (c == 1) x.y = v; else if (c == 2) x.z = v; else x.w = v;
  }
  return x;
}

// the real rows' share of gWe_ext[row, col .. col + 3] (EpiD::sp_rows)
__device__ inline float4 sparse_rows_quad(const EpiD& e, int row, int col) {
  if (row > e.sp_bias_row) return make_float4(0.f, 0.f, 0.f, 0.f);
  float4 s = *reinterpret_cast<const float4*>(e.sp_rows + (size_t)row * e.sp_ld + col);
  if (row == e.sp_bias_row)
    for (int q = 1; q < e.sp_bias_parts; ++q) {
      const float4 d = *reinterpret_cast<const float4*>(e.sp_rows + (size_t)(row + q) * e.sp_ld + col);
      s.x += d.x; s.y += d.y; s.z += d.z; s.w += d.w;
    }
  return s;
}

// theta - lr_t m / (sqrt(v) + eps), the last line of TF ApplyAdam, for every Adam kernel of the library (one definition: the fused
// epilogues, the stand-alone and the per-row kernels must agree bit for bit).  v_sqrt_f32 and v_rcp_f32 are 1 ulp each, so the step is
// within 2.5 ulp (3e-7 relative) of the correctly rounded quotient -- with |step| <= lr that is 1e-10 on parameters whose own fp32
// spacing is 1e-9 ... 1e-7, three orders inside the path's stated tolerance (1e-4 relative on the scores), and the explicit fma gives every call
// site the same rounding.  The IEEE sqrtf / divide sequences cost ~20 more vector instructions per element (2 us of the 46 us
// weight-gradient launch, profiles/r04_wgrad_stream.md); `make ADAM_IEEE=1` builds them.
__device__ __forceinline__ float adam_step(float x, float ma, float v) {
#ifdef GANMF_ADAM_IEEE
  return x - ma / (sqrtf(v) + ADAM_EPS);
#else
  return __builtin_fmaf(-ma, __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(v) + ADAM_EPS), x);
#endif
}

// TF ApplyAdam on one element (GANMF.py:104-105,138; training_ops ApplyAdam functor)
__device__ inline void adam_update(float g, float alpha, float reg, float& th, float& m, float& v, float& sq) {
  const float x = th;
  sq += x * x;
  const float gr = g + reg * x;
  m += (gr - m) * (1.f - ADAM_B1);
  v += (gr * gr - v) * (1.f - ADAM_B2);
  th = adam_step(x, m * alpha, v);
}

__device__ inline float epi_apply(const EpiD& e, float v, int row, int col, int ldc, const float* __restrict__ aux,
                                  float& sq) {
  switch (e.kind) {
    case EPI_SUB_AUX_SQ:
      if (aux) v -= aux[(size_t)row * e.ldaux + col];      // (aux == nullptr: the CSR path subtracted already, csr_quad)
      sq += v * v;
      break;
    case EPI_SUB_SCALED_AUX:
      v -= e.c * aux[(size_t)row * e.ldaux + col];
      break;
    case EPI_ROWSCALE:
      v *= e.rowscale[row];
      break;
    case EPI_G_DE: {
      const float d = e.ef[(size_t)row * ldc + col] - e.er[(size_t)row * ldc + col];
      v = e.c * v + e.cfm * d;
      sq += d * d;
      break;
    }
    case EPI_ACT:
      if (e.r1_u) v += e.r1_u[(size_t)row * e.r1_ld] * e.r1_w[col];
      v = act_apply(e.act, v);
      break;
    case EPI_MUL_ACTGRAD:
      v *= act_grad_out(e.act, aux[(size_t)row * e.ldaux + col]);
      break;
    default: break;
  }
  return v;
}

struct GemmP {
  const float* A;
  const float* B;
  float* C;
  int lda, ldb, ldc;
  int M, N, K;
  const float* zero_page;  // >= 4 KiB of zeros in device memory (out-of-range lanes read zero_page + 16 B * (tid & 255))
  int nsplit;              // >= 1; > 1: C is the slab, epilogue deferred to splitk_reduce_kernel
  int k_per_split;         // multiple of BK
  long long c_split_stride;
  int nbatch;              // >= 1; B is shared between batches
  long long a_batch_stride, c_batch_stride;
  EpiD epi;
  int tiles_m, tiles_n;
  // split-K finished inside the launch: the last workgroup to arrive at a tile (agent-scope
  // arrival counter) sums the slabs in split order and applies the epilogue into Cf
  unsigned* counters;      // [nbatch * tiles]; zero between launches (the reducer re-zeroes its word)
  float* Cf;               // final output [M, ldc] per batch
  long long cf_batch_stride;
  int c_pad_writable;           // columns N .. ldc-1 of C hold nothing the caller needs (gemm_persist.hpp writes zeros there)
  const int* a_gather;          // K-contiguous A only: row r of A is A + a_gather[r] * lda (embedding lookup folded into the
                                // operand fetch, GANMF.py:82); nullptr: row r is A + r * lda
  int a_gather_batch;           // batch bz reads its row list at a_gather + bz * a_gather_batch (one product per minibatch of a staged pass)
  float a_scale, b_scale;       // MFMA_F16: powers of two applied to the operands at conversion (0 = 1); acc *= 1 / (a_scale * b_scale)
  // Blocked tile order of the one-tile-per-workgroup kernels (0 = tm-fastest list order).  Plain products with many more
  // tiles than CUs (the scoring GEMM): the tile grid is cut into xb_m x xb_n = 8 rectangles, one per XCD, and a rectangle
  // is walked in bands of xb_band tile rows, M-innermost, so an XCD's L2 keeps its A band and streams its B panels once per
  // band (tile_coords below; speed only, any order is correct).
  int xb_m, xb_n, xb_band;
  int diag;                     // diagnostic builds only (make DIAG=1): timing experiments of the staged kernel's K loop
  unsigned long long* stamps;   // diagnostic builds only: four s_memrealtime stamps per workgroup (entry, first K-tile landed, K loop done, exit)
  // Operands split ahead of the launch (gemm_planes.hpp): the three bf16 pieces of A / B as row-major matrices of the fp32 operand's
  // geometry (same lda / ldb, in elements), piece q at base + q * stride.  Both set (and A K-contiguous, no a_gather): the 16-wave
  // split-bf16 plan runs its DMA form -- no split in the K loop, bit-identical results.  nullptr: the fp32 operand is split in the loop.
  const unsigned short* a_planes;
  const unsigned short* b_planes;
  long long a_pstride, b_pstride;
  int n_fastest;                // list order with the tile COLUMN fastest (default: tile row fastest).  For the fused-Adam weight-gradient
                                // products: workgroups that run at the same time then update neighbouring 256-byte segments of the same
                                // parameter rows, i.e. whole DRAM pages of theta / m / v instead of one segment per 4-15 KiB row
};

#define GANMF_WAIT_VMCNT(N) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory")

// One operand's share of a K-tile: R rows (M or N side) x BK k's, staged by NTHR threads.
template <int R, int BK, bool KM, int NTHR = 256>
struct Stage {
  static constexpr int F4 = R * BK / 4;          // float4 per tile
  static constexpr int NP = F4 / NTHR;           // glds per thread per tile
  static constexpr int SZ = R * BK;              // floats in LDS (unpadded)
  static constexpr int S = BK / 4;               // K-contig: 16-byte slots per row
  static constexpr int RC4 = R / 4;              // K-major: 16-byte slots per k-row
  static_assert(F4 % NTHR == 0 && NP >= 1, "tile too small for the workgroup");
  static_assert(BK == 32 || BK == 64, "BK must be 32 or 64");

  // rows that share a 64-dword LDS bank row differ in bit 0 (BK=32) or not at all (BK=64)
  __device__ static inline int swz(int row) { return (row / (64 / BK)) & (S - 1); }

  const float* zp;        // this lane's own line of the zero page
  const float* ptr[NP];   // per-lane source address of the next tile (KM: before validity select)
  int aux[NP];            // !KM: pointer increment per tile (0 for zero-page lanes); KM: k-row or -1

  __device__ inline void init(const float* __restrict__ base, int ld, int r0, int rlimit, int kbeg,
                              const float* zero, int tid, const int* __restrict__ gather = nullptr) {
    zero += (tid & 255) * 4;   // distinct lines per lane: no single-line hot spot
    zp = zero;
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      const int pos = j * NTHR + tid;
      if constexpr (!KM) {
        const int row = pos / S, slot = pos % S;
        const int c4 = slot ^ swz(row);
        const bool ok = (r0 + row) < rlimit;
        const int src_row = (ok && gather) ? gather[r0 + row] : r0 + row;
        ptr[j] = ok ? base + (size_t)src_row * ld + kbeg + 4 * c4 : zero;
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
  // one glds piece (1 KiB per wave) of the tile that starts kleft k's before the end of the K range
  template <int J>
  __device__ inline void issue_piece(float* s, int ld, int kleft, int wave) {
    const float* src;
    if constexpr (!KM) {
      src = kleft > 0 ? ptr[J] : zp;
      ptr[J] += aux[J];
    } else {
      src = (aux[J] >= 0 && aux[J] < kleft) ? ptr[J] : zp;
      ptr[J] += (size_t)BK * ld;
    }
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)(s + (J * NTHR + wave * 64) * 4),
                                     16, 0, 0);
  }
  // The same piece issued from inline asm (recipe: cdna guide, inline-asm section, `glds16_asm`).  hipcc models the
  // builtin as a FLAT instruction that may touch both VMEM and LDS; while one is pending, every lgkmcnt dependency is
  // waited for with lgkmcnt(0) (SIInsertWaitcnts "pending flat"), so the double-buffered fragment reads of the K loop lose
  // their lead: the reads issued for the NEXT chunk are waited for before THIS chunk's MFMAs.  An asm statement is
  // invisible to that pass: fragment waits become counted again; the vmcnt side is counted by hand in these kernels anyway.
  template <int J>
  __device__ inline void issue_piece_asm(float* s, int ld, int kleft, int wave) {
    const float* src;
    if constexpr (!KM) {
      src = kleft > 0 ? ptr[J] : zp;
      ptr[J] += aux[J];
    } else {
      src = (aux[J] >= 0 && aux[J] < kleft) ? ptr[J] : zp;
      ptr[J] += (size_t)BK * ld;
    }
    const unsigned dst = __builtin_amdgcn_readfirstlane(
        (unsigned)(size_t)(__attribute__((address_space(3))) void*)(s + (J * NTHR + wave * 64) * 4));
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
  }
  template <int J0, int J1>
  __device__ inline void issue_range_asm(float* s, int ld, int kleft, int wave) {
    if constexpr (J0 < J1) {
      issue_piece_asm<J0>(s, ld, kleft, wave);
      issue_range_asm<J0 + 1, J1>(s, ld, kleft, wave);
    }
  }
  template <int J0, int J1>
  __device__ inline void issue_range(float* s, int ld, int kleft, int wave) {
    if constexpr (J0 < J1) {
      issue_piece<J0>(s, ld, kleft, wave);
      issue_range<J0 + 1, J1>(s, ld, kleft, wave);
    }
  }
  __device__ inline void issue(float* s, int ld, int kleft, int wave) {
    issue_range<0, NP>(s, ld, kleft, wave);
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

__host__ __device__ inline int part_begin(int n, int parts, int i) { return (int)(((long long)n * i) / parts); }

// block id -> (tile row, tile column, K split, batch).  Default: list order, tm fastest, each XCD a contiguous range of the
// list.  GemmP::xb_m > 0: the list is re-ordered rectangle by rectangle (rectangle r = (r % xb_m, r / xb_m) of the tile grid),
// inside a rectangle band by band, inside a band M-innermost; XCD x still takes a contiguous range of the list, i.e. its own
// rectangle up to a few tiles of drift where the rectangles' sizes differ.  Bijective for every shape.
__device__ inline void tile_coords(const GemmP& p, int bid, int nblk, int& tm, int& tn, int& sp, int& bz) {
  int t = xcd_remap(bid, nblk);
  if (p.xb_m > 0) {
    sp = 0; bz = 0;
    int mb0 = 0, nb0 = 0, bm = 1, bn = 1;
    for (int r = 0; r < 8; ++r) {
      const int i = r % p.xb_m, j = r / p.xb_m;
      mb0 = part_begin(p.tiles_m, p.xb_m, i); bm = part_begin(p.tiles_m, p.xb_m, i + 1) - mb0;
      nb0 = part_begin(p.tiles_n, p.xb_n, j); bn = part_begin(p.tiles_n, p.xb_n, j + 1) - nb0;
      if (t < bm * bn) break;
      t -= bm * bn;
    }
    const int bh = min(max(p.xb_band, 1), bm);     // band height in tiles
    const int band = t / (bh * bn);
    const int r = t - band * bh * bn;
    const int h = min(bh, bm - band * bh);          // the last band of a rectangle may be shorter
    tm = mb0 + band * bh + r % h;
    tn = nb0 + r / h;
    return;
  }
  if (p.n_fastest) {
    tn = t % p.tiles_n; t /= p.tiles_n;
    tm = t % p.tiles_m; t /= p.tiles_m;
  } else {
    tm = t % p.tiles_m; t /= p.tiles_m;
    tn = t % p.tiles_n; t /= p.tiles_n;
  }
  sp = t % p.nsplit;
  bz = t / p.nsplit;
}

#ifndef GANMF_ADAM_HOIST
#define GANMF_ADAM_HOIST 1
#endif

// ---- epilogue shared by every GEMM kernel.  C/D layout of the 32x32 MFMA: col = lane & 31,
// row = (reg&3) + 8*(reg>>2) + 4*(lane>>5), i.e. a lane owns a column.  `smem` must hold BM*BN floats and be idle.
struct TileCoord { int tm, tn, sp, bz, m0, n0; };

// KG > 1: the workgroup has KG groups of four waves that each hold a partial sum of the SAME tile (they split the
// chunks of every K-tile between them, gemm_f32_mfma); group g stages its accumulators at smem + g * BM * BN and the row
// pass adds the KG images in group order (fixed order: bitwise reproducible).
// ADAM_OK: the instantiation carries the fused TF-Adam row pass (EPI_ADAM).  Every product that ends in it is a weight gradient, a TN
// product (gWd_ext, gWe_ext, gV, DisGANMF's layer gradients): the NT / NN kernels are built without it (gemm_dispatch rejects the
// combination), which takes the row pass, its twelve hoisted streams and the IEEE sqrt / divide sequences out of two thirds of the
// GEMM code objects.
// NIMG / STAGED (gemm_bf16w.hpp): the caller has staged NIMG partial images of the tile at smem + g * BM * BN itself (its waves are not the
// 2 x 2 grid this function files); the row pass below is shared.  A tile with fewer float4 than threads (64 x 32 on 1024) leaves the
// threads past it idle.
template <int BM, int BN, int TM, int TN, int KG = 1, bool ADAM_OK = true, int NIMG = KG, bool STAGED = false>
__device__ inline void gemm_epilogue(const GemmP& p, const f32x16 (&acc)[TM][TN], float* smem, const TileCoord& tc_) {
  constexpr int WM = BM / 2, WN = BN / 2;
  constexpr int NTHR = 256 * KG;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kg = wave >> 2;
  const int wr = (wave >> 1) & 1, wc = wave & 1;
  const int li = lane & 31, lh = lane >> 5;
  const int tm = tc_.tm, tn = tc_.tn, sp = tc_.sp, bz = tc_.bz, m0 = tc_.m0, n0 = tc_.n0;
  // Stored straight from registers that is one dword per lane per
  // instruction (measured ~2 TB/s chip-wide on 15-90 MB outputs); instead the tile is staged
  // through the now idle ring as a natural [BM][BN] image and written as whole rows, 16 B per lane.
  float* __restrict__ ct = smem + kg * (BM * BN);
  const bool deferred = p.nsplit > 1;
  const EpiD& e = p.epi;
  const bool csr0 = e.csr_indptr != nullptr && bz == 0 && e.kind == EPI_SUB_AUX_SQ;      // the real path's X stays CSR
  const float* __restrict__ aux = (e.aux && !csr0) ? e.aux + (size_t)bz * e.aux_batch_stride : nullptr;
  constexpr int C4 = BN / 4, RPP = NTHR / C4 < BM ? NTHR / C4 : BM;
  constexpr bool SPARE = NTHR / C4 > BM;      // more threads than float4 in the tile: rows past BM are nobody's
  static_assert(BM % RPP == 0, "row pass must cover the tile in whole steps");
  static_assert(!SPARE || !ADAM_OK, "the hoisted Adam streams assume every thread owns a row");
  const int tc = tid % C4, tr = tid / C4;
  const int col = n0 + tc * 4;
  // (round 6) A workgroup whose row pass is ONE step (the 16-wave kernels: 1024 threads on a 64 x 64 or 64 x 32 tile) requests its float4 of
  // the auxiliary matrix (decode: the input it subtracts; dF: Delta; DisGANMF's backward: the layer output) HERE, in front of the exchange of
  // the K groups' partial tiles through LDS and its barrier, instead of behind them: the in-kernel stamps show 1.5-2 us of a 3 us epilogue
  // waiting for that dependent fetch (profiles/r06_launch_fixed_part.md).  Same values, same arithmetic.
  float4 auxv = make_float4(0.f, 0.f, 0.f, 0.f);
  bool aux_pre = false;
  if constexpr (BM == RPP) {
    const int row = m0 + tr;
    if (!deferred && aux && (e.kind == EPI_SUB_AUX_SQ || e.kind == EPI_SUB_SCALED_AUX || e.kind == EPI_MUL_ACTGRAD) && row < p.M && col + 3 < p.N &&
        (!SPARE || tr < BM)) {
      auxv = *reinterpret_cast<const float4*>(aux + (size_t)row * e.ldaux + col);
      aux_pre = true;
    }
  }
  if constexpr (!STAGED) {
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int b = 0; b < TN; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          ct[(wr * WM + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * BN + wc * WN + b * 32 + li] = acc[a][b][r];
  }
  __syncthreads();

  float* __restrict__ C = p.C + (size_t)sp * p.c_split_stride + (size_t)bz * p.c_batch_stride;
  const bool sp_add = !(p.nsplit > 1) && e.sp_rows != nullptr;      // gWe_ext: the real rows' share comes from the CSC matrix
  float sq = 0.f;
  ct = smem;
  const bool adam = ADAM_OK && !deferred && e.kind == EPI_ADAM;
  const float alpha = adam ? *e.adam_alpha : 0.f;
  float* __restrict__ theta_out = e.adam_theta_out ? e.adam_theta_out : e.adam_theta;
#ifdef GANMF_PERSIST_DIAG_BUILD
  const bool inlaunch = deferred && p.counters != nullptr;      // in-launch split-K reduction: an experiment (measured level / slower)
#else
  constexpr bool inlaunch = false;
#endif
  const bool publish = inlaunch;
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  const __amdgpu_buffer_rsrc_t slab_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)C, (short)0, 0x7fffffff, 0x00020000);
#ifdef GANMF_PERSIST_DIAG_BUILD
  if (adam && (p.diag & 8)) return;      // timing only: the K phase without the Adam streams
#endif
  if (adam && GANMF_ADAM_HOIST) {
    // The tile is a gradient: parameter and moments are updated in place, the gradient is never stored.  theta / m / v of up to
    // four row steps are fetched BEFORE the first of them is written back: written as one loop the stores of step j and the
    // loads of step j + 1 go through the same pointers, so the compiler keeps them in order and the row pass becomes four
    // dependent load -> update -> store rounds of HBM latency each.
    // (JC row steps' streams requested ahead.  Rounds 2-4: four -- a lone workgroup's row pass then waits for HBM once instead of four
    // times; with six workgroups of the weight-gradient launch resident per CU the others' K loops cover that wait, and the twelve
    // requests per thread at once cost more than they hide: the launch 45.3 / 44.5 / 44.1 us with JC = 4 / 2 / 1, round 5)
    constexpr int J = BM / RPP, JC = 1;
    static_assert(J % JC == 0, "row steps in whole chunks");
    for (int j0 = 0; j0 < J; j0 += JC) {
      float4 t4[JC], m4[JC], v4[JC];
#pragma unroll
      for (int jj = 0; jj < JC; ++jj) {
        const int row = m0 + tr + (j0 + jj) * RPP;
        if (row < p.M && col + 3 < p.N) {
          const size_t off = (size_t)row * p.ldc + col;
          t4[jj] = *reinterpret_cast<const float4*>(e.adam_theta + off);
          m4[jj] = *reinterpret_cast<const float4*>(e.adam_m + off);
          v4[jj] = *reinterpret_cast<const float4*>(e.adam_v + off);
        }
      }
#pragma unroll
      for (int jj = 0; jj < JC; ++jj) {
        const int row_l = tr + (j0 + jj) * RPP, row = m0 + row_l;
        if (row < p.M && col < p.N) {
          float4 v = *reinterpret_cast<const float4*>(ct + row_l * BN + tc * 4);
#pragma unroll
          for (int g = 1; g < NIMG; ++g) {
            const float4 w = *reinterpret_cast<const float4*>(ct + g * (BM * BN) + row_l * BN + tc * 4);
            v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
          }
          if (sp_add) {
            const float4 x = sparse_rows_quad(e, row, col);
            v.x += x.x; v.y += x.y; v.z += x.z; v.w += x.w;
          }
          const size_t off = (size_t)row * p.ldc + col;
          if (col + 3 < p.N) {
            adam_update(v.x, alpha, e.adam_reg, t4[jj].x, m4[jj].x, v4[jj].x, sq);
            adam_update(v.y, alpha, e.adam_reg, t4[jj].y, m4[jj].y, v4[jj].y, sq);
            adam_update(v.z, alpha, e.adam_reg, t4[jj].z, m4[jj].z, v4[jj].z, sq);
            adam_update(v.w, alpha, e.adam_reg, t4[jj].w, m4[jj].w, v4[jj].w, sq);
            *reinterpret_cast<float4*>(theta_out + off) = t4[jj];
            *reinterpret_cast<float4*>(e.adam_m + off) = m4[jj];
            *reinterpret_cast<float4*>(e.adam_v + off) = v4[jj];
          } else {
            const float o[4] = {v.x, v.y, v.z, v.w};
            for (int q = 0; q < 4 && col + q < p.N; ++q) {
              float th = e.adam_theta[off + q];
              adam_update(o[q], alpha, e.adam_reg, th, e.adam_m[off + q], e.adam_v[off + q], sq);
              theta_out[off + q] = th;
            }
          }
        }
      }
    }
  } else
#pragma unroll 4
  for (int j = 0; j < BM / RPP; ++j) {
    const int row_l = tr + j * RPP, row = m0 + row_l;
    if (row < p.M && col < p.N && (!SPARE || row_l < BM)) {
      float4 v = *reinterpret_cast<const float4*>(ct + row_l * BN + tc * 4);
#pragma unroll
      for (int g = 1; g < NIMG; ++g) {
        const float4 w = *reinterpret_cast<const float4*>(ct + g * (BM * BN) + row_l * BN + tc * 4);
        v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
      }
      if (sp_add) {
        const float4 x = sparse_rows_quad(e, row, col);
        v.x += x.x; v.y += x.y; v.z += x.z; v.w += x.w;
      }
      if (csr0 && !deferred) {
        const float4 x = csr_quad(e, row, col);
        v.x -= x.x; v.y -= x.y; v.z -= x.z; v.w -= x.w;
      }
      float o[4] = {v.x, v.y, v.z, v.w};
      if (adam) {   // the tile is a gradient: update parameter and moments in place (never stored)
        const size_t off = (size_t)row * p.ldc + col;
        if (col + 3 < p.N) {
          float4 t4 = *reinterpret_cast<float4*>(e.adam_theta + off);
          float4 m4 = *reinterpret_cast<float4*>(e.adam_m + off);
          float4 v4 = *reinterpret_cast<float4*>(e.adam_v + off);
          adam_update(o[0], alpha, e.adam_reg, t4.x, m4.x, v4.x, sq);
          adam_update(o[1], alpha, e.adam_reg, t4.y, m4.y, v4.y, sq);
          adam_update(o[2], alpha, e.adam_reg, t4.z, m4.z, v4.z, sq);
          adam_update(o[3], alpha, e.adam_reg, t4.w, m4.w, v4.w, sq);
          *reinterpret_cast<float4*>(theta_out + off) = t4;
          *reinterpret_cast<float4*>(e.adam_m + off) = m4;
          *reinterpret_cast<float4*>(e.adam_v + off) = v4;
        } else {
          for (int q = 0; q < 4 && col + q < p.N; ++q) {
            float th = e.adam_theta[off + q];
            adam_update(o[q], alpha, e.adam_reg, th, e.adam_m[off + q], e.adam_v[off + q], sq);
            theta_out[off + q] = th;
          }
        }
        continue;
      }
      if (publish) {
        // slab tile handed to another workgroup inside this launch: WRITE-THROUGH (sc1) stores,
        // so no release fence is needed (cdna guide §6 Guideline 16, R1)
        const int boff = (int)(((size_t)row * p.ldc + col) * sizeof(float));
        if (col + 3 < p.N) {
          u32x4 bits;
          bits[0] = __float_as_uint(o[0]); bits[1] = __float_as_uint(o[1]);
          bits[2] = __float_as_uint(o[2]); bits[3] = __float_as_uint(o[3]);
          __builtin_amdgcn_raw_buffer_store_b128(bits, slab_rsrc, boff, 0, 16 /* sc1 */);
        } else {
          for (int q = 0; q < 4 && col + q < p.N; ++q)
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(o[q]), slab_rsrc, boff + 4 * q, 0, 16);
        }
      } else if (col + 3 < p.N) {
        if (aux_pre) {      // (the auxiliary values arrived while the partial tiles met in LDS: epi_apply's three cases on them)
          const float a4[4] = {auxv.x, auxv.y, auxv.z, auxv.w};
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            if (e.kind == EPI_SUB_AUX_SQ) { o[q] -= a4[q]; sq += o[q] * o[q]; }
            else if (e.kind == EPI_SUB_SCALED_AUX) o[q] -= e.c * a4[q];
            else o[q] *= act_grad_out(e.act, a4[q]);
          }
        } else if (!deferred) {
#pragma unroll
          for (int q = 0; q < 4; ++q) o[q] = epi_apply(e, o[q], row, col + q, p.ldc, aux, sq);
        }
        *reinterpret_cast<float4*>(C + (size_t)row * p.ldc + col) = make_float4(o[0], o[1], o[2], o[3]);
      } else {
        for (int q = 0; q < 4 && col + q < p.N; ++q)
          C[(size_t)row * p.ldc + col + q] = deferred ? o[q] : epi_apply(e, o[q], row, col + q, p.ldc, aux, sq);
      }
    }
  }
  bool write_sq = !deferred && e.sq_partials;
  if (inlaunch) {
    // ---- in-launch split-K reduction (cdna guide §5 "In-launch split-K reduction", sc1 form):
    // write-through slab stores -> every wave drains vmcnt -> workgroup barrier -> lane 0 relaxed
    // agent fetch_add; the workgroup that draws nsplit-1 acquires (L1 invalidate) and reduces.
    // Correct for any placement of a tile's slices over XCDs; the sum runs in split order.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int* flag = reinterpret_cast<int*>(smem);
    if (tid == 0) {
      unsigned* cnt = p.counters + (size_t)bz * p.tiles_m * p.tiles_n + tn * p.tiles_m + tm;
      const unsigned prev = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int last = prev == (unsigned)(p.nsplit - 1);
      if (last) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      *flag = last;
    }
    __syncthreads();
    if (!*flag) return;
    const float* __restrict__ slab = p.C + (size_t)bz * p.c_batch_stride;
    float* __restrict__ Cf = p.Cf + (size_t)bz * p.cf_batch_stride;
#pragma unroll 2
    for (int j = 0; j < BM / RPP; ++j) {
      const int row = m0 + tr + j * RPP;
      if (row < p.M && col < p.N && (!SPARE || tr + j * RPP < BM)) {
        const size_t off = (size_t)row * p.ldc + col;
        float4 s4 = *reinterpret_cast<const float4*>(slab + off);
#pragma unroll 4
        for (int k = 1; k < p.nsplit; ++k) {
          const float4 q = *reinterpret_cast<const float4*>(slab + (size_t)k * p.c_split_stride + off);
          s4.x += q.x; s4.y += q.y; s4.z += q.z; s4.w += q.w;
        }
        if (csr0) {
          const float4 x = csr_quad(e, row, col);
          s4.x -= x.x; s4.y -= x.y; s4.z -= x.z; s4.w -= x.w;
        }
        float o[4] = {s4.x, s4.y, s4.z, s4.w};
        if (col + 3 < p.N) {
#pragma unroll
          for (int q = 0; q < 4; ++q) o[q] = epi_apply(e, o[q], row, col + q, p.ldc, aux, sq);
          *reinterpret_cast<float4*>(Cf + off) = make_float4(o[0], o[1], o[2], o[3]);
        } else {
          for (int q = 0; q < 4 && col + q < p.N; ++q) Cf[off + q] = epi_apply(e, o[q], row, col + q, p.ldc, aux, sq);
        }
      }
    }
    write_sq = e.sq_partials != nullptr;
  }
  if (write_sq) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
    __syncthreads();   // every wave is done reading the staged tile / the flag
    if (lane == 0) smem[wave] = sq;
    __syncthreads();
    if (tid == 0) {
      float t = (smem[0] + smem[1]) + (smem[2] + smem[3]);
#pragma unroll
      for (int g = 1; g < KG; ++g) t += (smem[4 * g] + smem[4 * g + 1]) + (smem[4 * g + 2] + smem[4 * g + 3]);
      if (e.sq_m_half > 0) e.sq_partials[(size_t)(tm / e.sq_m_half) * (p.tiles_n * e.sq_m_half) + tn * e.sq_m_half + tm % e.sq_m_half] = t;
      else e.sq_partials[(size_t)bz * e.sq_stride + tn * p.tiles_m + tm] = t;
    }
  }
}

// KG (1, 2 or 4) groups of four waves share ONE output tile: the groups stage the K-tiles together (every wave issues
// its share of the LDS-DMA pieces) and split the 8-wide chunks of each K-tile between them, chunk c to group c % KG, so a
// CU that holds a single workgroup still has KG waves per SIMD -- one wave's fragment reads, piece issue and barrier
// waits run under another's MFMAs (a single in-order wave per SIMD loses 15-30 % of the MFMA rate to them:
// profiles/README.md).  The partial sums meet in the epilogue through LDS.
// The body is a device function of (block index, blocks of this GEMM) so that one launch can carry several independent
// pieces of work (gemm_multi.hpp); `smem` is the launch's only LDS object, NS ring slots of BM*BK + BN*BK floats.
template <int BM, int BN, int BK, int NS, bool AKM, bool BKM, int KG = 1>
__device__ __forceinline__ void gemm_f32_body(const GemmP& p, const int bid, const int nblk, float* __restrict__ smem) {
  constexpr int NTHR = 256 * KG;
  using SA = Stage<BM, BK, AKM, NTHR>;
  using SB = Stage<BN, BK, BKM, NTHR>;
  constexpr int WM = BM / 2, WN = BN / 2;
  constexpr int TM = WM / 32, TN = WN / 32;
  static_assert(TM >= 1 && TN >= 1, "wave tile must hold at least one 32x32 MFMA block");
  static_assert(NS >= 2 && NS <= 4, "ring depth");
  constexpr int BUF = SA::SZ + SB::SZ;            // ring slot b: A at smem + b*BUF, B right behind
  constexpr int LOADS = SA::NP + SB::NP;          // glds per wave per tile

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kg = wave >> 2;                       // K group of this wave
  const int wr = (wave >> 1) & 1, wc = wave & 1;
  const int li = lane & 31, lh = lane >> 5;

  int tm, tn, sp, bz;
  tile_coords(p, bid, nblk, tm, tn, sp, bz);
#ifdef GANMF_PERSIST_DIAG_BUILD
#define GANMF_GEMM_STAMP(i) do { if (p.stamps && tid == 0) p.stamps[(size_t)bid * 4 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define GANMF_GEMM_STAMP(i) do { } while (0)
#endif
  GANMF_GEMM_STAMP(0);

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
  la.init(p.A + (size_t)bz * p.a_batch_stride, p.lda, m0, p.M, kbeg, p.zero_page, tid,
          (AKM || !p.a_gather) ? nullptr : p.a_gather + (size_t)bz * p.a_gather_batch);
  lb.init(p.B, p.ldb, n0, p.N, kbeg, p.zero_page, tid);

  constexpr int NC = BK / 8 / KG;   // 8-wide k chunks per tile and K group (even: local chunk c uses fragment set c & 1)
  static_assert((BK / 8) % KG == 0 && NC >= 2 && NC % 2 == 0, "chunks per K group");
  float4 fa[2][TM], fb[2][TN];
  auto load_frags = [&](int set, const float* __restrict__ tile, int cl) {
    const int c = cl * KG + kg;     // chunk of the K-tile behind this group's local chunk cl
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

  // prologue: tiles 0 .. min(NS, nt)-1 in flight.  No loads are issued past the K range (a dummy tile
  // makes every lane of every workgroup read the same zero-page line: an L2 hot spot that cost
  // 20-40 us on the short-K GEMMs); the tail therefore waits with vmcnt(0) instead of the counted wait.
  int kleft = kend - kbeg;   // k's remaining from the next tile to issue
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    if (kleft > 0) {
      la.issue(smem + s * BUF, p.lda, kleft, wave);
      lb.issue(smem + s * BUF + SA::SZ, p.ldb, kleft, wave);
    }
    kleft -= BK;
  }
  if (nt >= NS) GANMF_WAIT_VMCNT((NS - 1) * LOADS);   // tile 0 of this wave has landed ...
  else GANMF_WAIT_VMCNT(0);
  __builtin_amdgcn_s_barrier();                        // ... and of every other wave
  GANMF_GEMM_STAMP(1);
  load_frags(0, smem, 0);

  // The refill of a freed ring slot is spread over the chunks of the FOLLOWING tile, LOADS / NC glds per
  // chunk: a single wave per SIMD issues in order, so the ~60 address/issue instructions of a whole
  // tile in one block stall the MFMA pipe at every tile boundary; one or two pieces per chunk hide in
  // the MFMA gaps.  (All pieces of a tile are still issued between two boundary waits, so the counted
  // vmcnt bookkeeping is unchanged.)
  constexpr int PPC = (LOADS + NC - 1) / NC;      // pieces per chunk
  int slot = 0;         // ring slot of tile `it`
  int pend_slot = 0;    // slot being refilled during this tile (freed at the previous boundary)
  int pend_kleft = 0;   // <= 0: nothing to refill
  auto refill_chunk = [&](auto cc) {
    constexpr int c = decltype(cc)::value;
    if (pend_kleft > 0) {
      float* base = smem + pend_slot * BUF;
      constexpr int q0 = c * PPC < LOADS ? c * PPC : LOADS, q1 = (c + 1) * PPC < LOADS ? (c + 1) * PPC : LOADS;
      // pieces [0, NP_A) belong to A, [NP_A, LOADS) to B
      constexpr int a0 = q0 < SA::NP ? q0 : SA::NP, a1 = q1 < SA::NP ? q1 : SA::NP;
      constexpr int b0 = q0 > SA::NP ? q0 - SA::NP : 0, b1 = q1 > SA::NP ? q1 - SA::NP : 0;
      la.template issue_range<a0, a1>(base, p.lda, pend_kleft, wave);
      lb.template issue_range<b0, b1>(base + SA::SZ, p.ldb, pend_kleft, wave);
    }
  };
  // Two loops, not one loop with a conditional wait: tiles that still have NS-1 younger tiles behind them wait
  // with the counted vmcnt, the last NS-1 tiles drain with vmcnt(0).
  auto tile = [&](auto tail_tag) {
    constexpr bool TAIL = decltype(tail_tag)::value;
    const float* __restrict__ cur = smem + slot * BUF;
    const int nslot = (slot + 1 == NS) ? 0 : slot + 1;
    static_for<0, NC>([&](auto cc) {
      constexpr int c = decltype(cc)::value;
      if constexpr (c + 1 < NC) {
        load_frags((c + 1) & 1, cur, c + 1);     // next chunk's fragments under this chunk's MFMAs
        refill_chunk(cc);
        __builtin_amdgcn_sched_barrier(0);       // keep the reads ahead of the MFMAs (hipcc sinks them)
      } else {
        // Last chunk: its fragments are in registers once lgkmcnt drains, so this wave no longer
        // reads slot `slot`.  Tile it+1 has landed when at most NS-2 younger tiles are outstanding
        // (tiles it+2 .. it+NS-1 exist only while it+NS-1 < nt; in the tail nothing younger is in
        // flight and the wait is vmcnt(0)); after the barrier that holds for every wave and slot
        // `slot` is free: tile it+NS is refilled into it during the next tile.
        refill_chunk(cc);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if constexpr (TAIL) GANMF_WAIT_VMCNT(0);
        else GANMF_WAIT_VMCNT((NS - 2) * LOADS);
        __builtin_amdgcn_s_barrier();
        pend_slot = slot;
        pend_kleft = kleft;
        kleft -= BK;
        load_frags(0, smem + nslot * BUF, 0);    // first fragments of tile it+1 under the last MFMAs
        __builtin_amdgcn_sched_barrier(0);
      }
      mfmas(c & 1);
    });
    slot = nslot;
  };
  int it = 0;
  for (; it + NS - 1 < nt; ++it) tile(std::false_type{});
  for (; it < nt; ++it) tile(std::true_type{});
  // nothing is in flight any more; LDS reads must be done before the ring is reused as C staging
  GANMF_WAIT_VMCNT(0);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  GANMF_GEMM_STAMP(2);
  static_assert(KG * BM * BN <= NS * BUF, "the ring must hold the KG staged partial tiles");
  gemm_epilogue<BM, BN, TM, TN, KG, AKM && BKM>(p, acc, smem, TileCoord{tm, tn, sp, bz, m0, n0});
#ifdef GANMF_PERSIST_DIAG_BUILD
  if (p.stamps) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
#endif
  GANMF_GEMM_STAMP(3);
}

template <int BM, int BN, int BK, int NS, bool AKM, bool BKM, int KG = 1>
__global__ __launch_bounds__(256 * KG) void gemm_f32_mfma(const GemmP p) {
  __shared__ __attribute__((aligned(16))) float smem[NS * (BM + BN) * BK];   // the ONLY LDS object (cdna guide §5 item 4a)
  gemm_f32_body<BM, BN, BK, NS, AKM, BKM, KG>(p, (int)blockIdx.x, (int)gridDim.x, smem);
}

typedef unsigned short bf16raw;
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

// ---- planes: x = hi + mid + lo exactly (split_bf16x3), three row-major bf16 matrices of the fp32 matrix's geometry -----------
struct PlaneRef {
  bf16raw* p;             // piece q at p + q * pstride
  long long pstride;      // elements
};

__device__ inline void planes_store4(const PlaneRef& pl, size_t off, float x, float y, float z, float w) {
  unsigned h0, m0, l0, h1, m1, l1;
  split_bf16x3(x, y, h0, m0, l0);
  split_bf16x3(z, w, h1, m1, l1);
  *reinterpret_cast<uint2*>(pl.p + off) = make_uint2(h0, h1);
  *reinterpret_cast<uint2*>(pl.p + pl.pstride + off) = make_uint2(m0, m1);
  *reinterpret_cast<uint2*>(pl.p + 2 * pl.pstride + off) = make_uint2(l0, l1);
}
__device__ inline void planes_store1(const PlaneRef& pl, size_t off, float x) {
  unsigned h, m, l;
  split_bf16x3(x, 0.f, h, m, l);
  pl.p[off] = (bf16raw)(h & 0xffffu);
  pl.p[pl.pstride + off] = (bf16raw)(m & 0xffffu);
  pl.p[2 * pl.pstride + off] = (bf16raw)(l & 0xffffu);
}


// Reduce split-K slabs and apply the deferred epilogue: out[m,n] = epi(sum_s part[s][m,n]).
// grid = (gx, nbatch); columns >= N are never written (ones / pad columns keep their values).
struct RedP {
  const float* part;
  long long split_stride;
  int nsplit;
  float* out;
  int ld;       // shared by part / out
  int M, N;
  long long batch_stride;   // of part and out
  EpiD epi;
  int sc1 = 0;              // 1: the sums are handed to OTHER workgroups of the same launch (wgrad_seam_kernel): write-through (sc1) 16-byte stores
};

// Slab groups of the reduce: a small output behind a deep split (the reference's defaults on LastFM: 64 x 32 outputs, K = 17 632,
// 250 slabs) leaves two workgroups summing 250 slabs one after the other (9.4 us).  Then G = 16 threads share an output element:
// thread group g sums slabs g, g + G, ... and the G partial sums meet in LDS, added in group order.  G depends on the shape
// only (never on the grid), so every launch that carries this reduce forms the same sums.
__host__ __device__ inline int reduce_groups(long long total4, int nsplit) {
  return (nsplit >= 64 && total4 <= 8192) ? 16 : 1;      // (shallower splits finish inside the launch floor anyway)
}

// (bx of nbx blocks of NT threads walk the elements of batch bz; `red` = 4 + 4 * NT floats of LDS)
template <int NT = 256>
__device__ __forceinline__ void splitk_reduce_body(const RedP& p, const int bx, const int nbx, const int bz, float* __restrict__ red) {
  const int n4 = (p.N + 3) >> 2;
  const long long total = (long long)p.M * n4;
  const float* __restrict__ part = p.part + (size_t)bz * p.batch_stride;
  float* __restrict__ out = p.out + (size_t)bz * p.batch_stride;
  const EpiD& e = p.epi;
  const bool csr0 = e.csr_indptr != nullptr && bz == 0 && e.kind == EPI_SUB_AUX_SQ;
  const float* __restrict__ aux = (e.aux && !csr0) ? e.aux + (size_t)bz * e.aux_batch_stride : nullptr;
  float sq = 0.f;
  auto finish = [&](float4 s, int m, int c, size_t off) {
    if (csr0) {
      const float4 x = csr_quad(e, m, c);
      s.x -= x.x; s.y -= x.y; s.z -= x.z; s.w -= x.w;
    }
    float o[4] = {s.x, s.y, s.z, s.w};
    if (c + 3 < p.N) {
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = epi_apply(e, o[j], m, c + j, p.ld, aux, sq);
      if (p.sc1) {      // (cdna guide section 6, Guideline 16 R1: payload stored write-through, the storing waves drain before the arrival is counted)
        typedef unsigned u32x4s __attribute__((ext_vector_type(4)));
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)out, (short)0, 0x7fffffff, 0x00020000);
        u32x4s bits;
        bits[0] = __float_as_uint(o[0]); bits[1] = __float_as_uint(o[1]); bits[2] = __float_as_uint(o[2]); bits[3] = __float_as_uint(o[3]);
        __builtin_amdgcn_raw_buffer_store_b128(bits, rs, (int)(off * sizeof(float)), 0, 16 /* sc1 */);
      } else
      *reinterpret_cast<float4*>(out + off) = make_float4(o[0], o[1], o[2], o[3]);
      if (e.planes) planes_store4(PlaneRef{e.planes, e.plane_stride}, off, o[0], o[1], o[2], o[3]);
    } else {
      for (int j = 0; j < 4 && c + j < p.N; ++j) {
        const float r = epi_apply(e, o[j], m, c + j, p.ld, aux, sq);
        if (p.sc1) {
          const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)out, (short)0, 0x7fffffff, 0x00020000);
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(r), rs, (int)((off + j) * sizeof(float)), 0, 16);
        } else
        out[off + j] = r;
        if (e.planes) planes_store1(PlaneRef{e.planes, e.plane_stride}, off + j, r);
      }
    }
  };
  const int G = reduce_groups(total, p.nsplit);
  if (G > 1) {
    float4* const meet = reinterpret_cast<float4*>(red + 4);      // [G][NT / G]
    const int epb = NT / G, g = threadIdx.x / epb, el = threadIdx.x % epb;
    for (long long base = (long long)bx * epb; base < total; base += (long long)nbx * epb) {      // uniform per workgroup
      const long long idx = base + el;
      const bool live = idx < total;
      const int m = live ? (int)(idx / n4) : 0, c = live ? (int)(idx % n4) * 4 : 0;
      const size_t off = (size_t)m * p.ld + c;
      float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
      if (live) {
#pragma unroll 8
        for (int k = g; k < p.nsplit; k += G) {
          const float4 q = *reinterpret_cast<const float4*>(part + (size_t)k * p.split_stride + off);
          s.x += q.x; s.y += q.y; s.z += q.z; s.w += q.w;
        }
      }
      meet[threadIdx.x] = s;
      __syncthreads();
      if (g == 0 && live) {
        for (int q = 1; q < G; ++q) {
          const float4 t = meet[q * epb + el];
          s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
        }
        finish(s, m, c, off);
      }
      __syncthreads();
    }
  } else {
    for (long long idx = (long long)bx * NT + threadIdx.x; idx < total; idx += (long long)nbx * NT) {
      const int m = (int)(idx / n4), c = (int)(idx % n4) * 4;
      const size_t off = (size_t)m * p.ld + c;
      float4 s = *reinterpret_cast<const float4*>(part + off);
#pragma unroll 8
      for (int k = 1; k < p.nsplit; ++k) {   // independent loads: keep several slabs in flight
        const float4 q = *reinterpret_cast<const float4*>(part + (size_t)k * p.split_stride + off);
        s.x += q.x; s.y += q.y; s.z += q.z; s.w += q.w;
      }
      finish(s, m, c, off);
    }
  }
  if (e.sq_partials) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sq;
    __syncthreads();
    if (threadIdx.x == 0) {
      float t = (red[0] + red[1]) + (red[2] + red[3]);
#pragma unroll
      for (int w = 4; w < NT / 64; w += 4) t += (red[w] + red[w + 1]) + (red[w + 2] + red[w + 3]);
      e.sq_partials[(size_t)bz * e.sq_stride + bx] = t;
    }
  }
}

__global__ __launch_bounds__(256) void splitk_reduce_kernel(const RedP p) {
  __shared__ __attribute__((aligned(16))) float red[4 + 4 * 256];
  splitk_reduce_body(p, (int)blockIdx.x, (int)gridDim.x, (int)blockIdx.y, red);
}

// ---- host side: plan (tile, ring depth, split-K) and launch ------------------------------------
// Tile configurations: {128x128, BK 32} and {64x64, BK 64}; both have 32 KiB ring slots.
// Ring depth 2 (64 KiB, two workgroups per CU) or 3 (96 KiB, one per CU).
constexpr int GEMM_K_ALIGN = 64;   // split-K slices are multiples of this (>= every BK)
constexpr int GEMM_CUS = 256;
constexpr int GEMM_RED_GRID = 512;

struct GemmPlan {
  int tile = 128, ring = 2, nsplit = 1, kps = 0;
  int mode = MFMA_F32;       // MfmaMode of the K loop
  int tiles_m = 0, tiles_n = 0;
  int persist = 0;           // 1: gemm_persist.hpp (one workgroup per CU walks a list of output tiles)
  int bk = 0;                // staged bf16 kernels, 64 x 64 tiles: 32 = half-depth K-tiles (24 KiB of LDS: six workgroups per CU)
  int kg = 1;                // fp32 ring kernel, 64 x 64 tiles: K groups of four waves per workgroup (1, 2 or 4)
  int sq_count = 0;          // sq partial entries per batch this plan produces
  int tile_order = 0;        // GemmTune::tile_order
  int skinny = 0;            // > 0: the streaming kernel for K <= 64 (gemm_skinny.hpp), this many rows per wave
  int skinny_n = 0;          // 1: the streaming kernel for N <= 32 behind a long K (gemm_skinny.hpp, gemm_skinny_n_kernel)
  int wide32 = 0;            // 1: the 16-wave split-bf16 loop on 64 x 32 tiles with 128-deep K-tiles, unsplit (gemm_bf16w.hpp)
  double est_us = 0;
};

struct GemmTune {            // overrides (0 = automatic), settable from the environment
  int tile = 0, ring = 0, nsplit = 0;
  int mode = MFMA_AUTO;      // MfmaMode of every plan; MFMA_AUTO: chosen per GEMM by gemm_plan
  int persist = -1;          // persistent tile-walking kernel: -1 automatic, 0 never, 1 whenever the GEMM is eligible
  int bk = 0;                // GemmPlan::bk
  int kg = 0;                // GemmPlan::kg (0 = automatic)
  int tile_order = 0;        // one-tile-per-workgroup kernels: 0 list order (default), 1 XCD-blocked order (GemmP::xb_m) when the
                             // product has at least three tiles per CU, 2 blocked order for every unsplit, unbatched product (tests).
                             // Measured on the split-bf16 scoring GEMM (6040 x 3706 x 250): operand fetch 150 -> 51 MB per launch,
                             // time 93 -> 99 us -- the L2 misses it removes were MALL hits that the kernel was not waiting on, and
                             // the list order keeps the eight XCDs on the same A rows at the same time.  Kept as an option.
};

inline void split_plan(int K, int want, int& nsplit, int& kps) {
  int chunks = (K + GEMM_K_ALIGN - 1) / GEMM_K_ALIGN;
  if (want < 1) want = 1;
  if (want > chunks) want = chunks;
  int cps = (chunks + want - 1) / want;
  kps = cps * GEMM_K_ALIGN;
  nsplit = (chunks + cps - 1) / cps;
}

// Cost model (cycles at ~2.1 GHz under fp32 MFMA load).  Every GEMM of the step is small
// (<= 1.9 GFLOP, >= 12 us at the MFMA roof), so the plan is about filling 256 CUs evenly and
// keeping each workgroup's serial K-walk short; split-K pays a slab round trip + one launch.
// Tile shapes of the fp32 ring kernel: code -> BM x BN x BK, measured cycles per K-tile and fixed cycles per workgroup.
// (A tall 128 x 64 x 64 tile -- 48 KiB slots, 25 % fewer operand bytes per FLOP than 64 x 64 x 64 -- was built and measured
// in round 2: 5900 cycles per 4096-cycle K-tile, the same 70 % as the 64 x 64 tile, and its coarser K split costs more
// slabs: 31.9 vs 28.7 us on the encode GEMM.  Not kept.)
struct TileShape { int code, bm, bn, bk; double cyc_tile, fixed; };
constexpr TileShape kTileShapes[2] = {
    {128, 128, 128, 32, 4800.0, 9000.0},   // measured 2.4 us per K-tile
    {64, 64, 64, 64, 2900.0, 6000.0},      // measured 1.25 us per K-tile: 16 B/clk/CU through L2 -> LDS
};

inline GemmPlan gemm_plan(int M, int N, int K, int nbatch, bool wants_sq, const GemmTune& tune, bool no_split = false) {
  GemmPlan best;
  best.est_us = 1e30;
  const double gflop = 2.0 * M * (double)N * K * nbatch * 1e-9;
  for (const TileShape& ts : kTileShapes) {
    const int tile = ts.code;
    if (tune.tile && tune.tile != tile) continue;
    // an output at most 64 rows or columns wide fills at most half of a 128 x 128 tile: twice the padded MFMA work and a row
    // pass of 16 dependent load -> Adam -> store rounds on a handful of lanes (gV at the reference's default batch 32, k = 10:
    // 36 us on 128-tiles, 10 us on 64-tiles) -- the model below only sees the workgroup count
    if (!tune.tile && tile == 128 && (M <= 64 || N <= 64)) continue;
    const int tm = (M + ts.bm - 1) / ts.bm, tn = (N + ts.bn - 1) / ts.bn;
    const long long T = (long long)tm * tn * nbatch;
    const int bk = ts.bk;
    const double cyc_tile = ts.cyc_tile;
    const int max_split = no_split ? 1 : std::max(1, (K + GEMM_K_ALIGN - 1) / GEMM_K_ALIGN);
    for (int want = 1; want <= max_split; want = want < 8 ? want + 1 : want + (want + 3) / 4) {      // 1 .. 8, 10, 13, 17, ...
      if (tune.nsplit) want = std::min(tune.nsplit, max_split);   // forced: evaluate exactly this one
      int ns, kps;
      split_plan(K, want, ns, kps);
      const long long wgs = T * ns;
      const double k_tiles = std::ceil((double)std::min(kps, K) / bk);
      const double per_wg = ts.fixed + k_tiles * cyc_tile;   // prologue + epilogue + K walk (K sweep, MI355X)
      const double rounds = std::ceil((double)wgs / GEMM_CUS);
      double us = rounds * per_wg / 2100.0;
      if (ns > 1) us += 1.5 + (double)(ns + 1) * M * N * nbatch * 4.0 / 3.0e6;   // in-launch reduce tail + slab traffic at ~3 TB/s
      if (us < best.est_us) {
        best.est_us = us; best.tile = tile; best.nsplit = ns; best.kps = kps; best.tiles_m = tm; best.tiles_n = tn;
      }
      if (tune.nsplit) break;
    }
  }
  const long long wgs = (long long)best.tiles_m * best.tiles_n * nbatch * best.nsplit;
  best.ring = tune.ring ? tune.ring : (wgs > GEMM_CUS ? 2 : 3);
  if (best.nsplit == 1) best.kps = ((K + GEMM_K_ALIGN - 1) / GEMM_K_ALIGN) * GEMM_K_ALIGN;
  // MFMA_AUTO: the staged split-bf16 kernel has the better throughput per CU once two or more workgroups share a
  // CU (K sweeps on MI355X, 993x3706xK: 1.28 vs 1.41 us per 64x64x64 workgroup-tile; 6040x3706x250: 97 vs 141 us;
  // 4096^3: 172 vs 120 TFLOP/s), the fp32 ring kernel the shorter latency when a workgroup has its CU to itself
  // (256x3706x2048 on 232 workgroups: 48 vs 57 us).  Every K-heavy 128/256-row GEMM of the C2 step is planned onto
  // <= 256 workgroups and stays fp32.  GEMMs of >= 6 GFLOP (the C4-sized steps: K or N = 50 000) run long enough
  // per workgroup that the split-bf16 loop wins on any grid (C4 shard: 955 vs 846 steps/s with every GEMM on it).
  // (128 x 128 tiles have no 16-wave form: from 1.5 workgroups per CU on the staged kernel already leads -- generator GEMM at
  // C4 width, 128 x 50 000 x 250 on 391 workgroups: 49.5 -> 34.4 us)
  const bool many = wgs >= 2 * GEMM_CUS || (best.tile == 128 && 2 * wgs >= 3 * GEMM_CUS);
  best.mode = tune.mode != MFMA_AUTO ? tune.mode : (many || gflop >= 6.0 ? MFMA_BF16X3 : MFMA_F32);
  best.bk = tune.bk;
  best.tile_order = tune.tile_order;
  // K groups: a 64 x 64 workgroup that has its CU to itself (ring 3: 96 KiB) runs 16 waves, four per SIMD (C2 step:
  // 6130 -> 6660 steps/s with two per SIMD, 6760-6830 with four); co-resident ring-2 workgroups already interleave
  best.kg = (best.tile == 64 && best.mode == MFMA_F32 && best.ring <= 3) ? (tune.kg ? tune.kg : (best.ring == 3 ? 4 : 1)) : 1;
  if (best.kg == 4 && best.ring != 3) best.kg = 2;
#ifndef GANMF_PERSIST_DIAG_BUILD
  // the product build carries two fp32 64 x 64 points: one K group on a 2-slot ring (co-resident workgroups) and four K groups on a 3-slot
  // ring (a CU to itself); overrides that name another point (GANMF_TUNE kg = 2, ring = 4, ring = 3 with kg = 1) land on the nearest one
  if (best.tile == 64 && best.mode == MFMA_F32) {
    if (best.kg >= 2 && best.ring >= 3) { best.kg = 4; best.ring = 3; }
    else { best.kg = 1; best.ring = 2; }
  }
#endif
  best.sq_count = wants_sq ? best.tiles_m * best.tiles_n : 0;   // (separate reduce kernel: GEMM_RED_GRID, set by gemm_run)
  return best;
}

inline size_t gemm_slab_elems(const GemmPlan& pl, int M, int ldc, int nbatch) {
  return pl.nsplit > 1 ? (size_t)pl.nsplit * nbatch * M * ldc : 0;
}

template <int BM, int BN, int BK, int NS, int KG = 1>
inline hipError_t gemm_launch_t(hipStream_t st, const GemmP& p, bool akm, bool bkm) {
  const int grid = p.tiles_m * p.tiles_n * p.nsplit * p.nbatch;
  if (grid <= 0) return hipSuccess;
  if (!akm && !bkm) GANMF_LAUNCH((gemm_f32_mfma<BM, BN, BK, NS, false, false, KG>), dim3(grid), dim3(256 * KG), 0, st, p);
  else if (!akm && bkm) GANMF_LAUNCH((gemm_f32_mfma<BM, BN, BK, NS, false, true, KG>), dim3(grid), dim3(256 * KG), 0, st, p);
  else if (akm && bkm) GANMF_LAUNCH((gemm_f32_mfma<BM, BN, BK, NS, true, true, KG>), dim3(grid), dim3(256 * KG), 0, st, p);
  else return hipErrorInvalidValue;  // TT is not needed by the GANMF step
  return hipGetLastError();
}

struct GemmPlan;
inline hipError_t gemm_dispatch(hipStream_t st, const GemmP& p, bool akm, bool bkm, const GemmPlan& pl);

// Logical GEMM: C[bz] = epi(op(A[bz]) . op(B)).  `p` carries the operands, shapes, batch strides
// and the epilogue; C/ldc/c_batch_stride describe the FINAL output.  slab: workspace for split-K.
inline hipError_t gemm_run(hipStream_t st, GemmP p, bool akm, bool bkm, const GemmPlan& pl, float* slab,
                           size_t slab_elems, unsigned* counters = nullptr, size_t n_counters = 0, int red_blocks = GEMM_RED_GRID) {
  if (p.nbatch < 1) p.nbatch = 1;
  if (!p.zero_page || (p.lda % LD_ALIGN) || (p.ldb % LD_ALIGN)) return hipErrorInvalidValue;
  p.tiles_m = pl.tiles_m; p.tiles_n = pl.tiles_n;
  p.nsplit = pl.nsplit; p.k_per_split = pl.kps;
  p.epi.sq_stride = pl.sq_count;
  p.counters = nullptr;
  const bool in_launch = pl.nsplit > 1 && counters && (size_t)pl.tiles_m * pl.tiles_n * p.nbatch <= n_counters;
  RedP r{};
  if (pl.nsplit > 1) {
    if (gemm_slab_elems(pl, p.M, p.ldc, p.nbatch) > slab_elems) return hipErrorOutOfMemory;
    r.part = slab; r.nsplit = pl.nsplit; r.out = p.C; r.ld = p.ldc; r.M = p.M; r.N = p.N;
    r.batch_stride = p.c_batch_stride; r.epi = p.epi;
    // slab layout [split][batch][M, ldc]
    r.split_stride = (long long)p.nbatch * p.M * p.ldc;
    if (p.nbatch > 1 && p.c_batch_stride != (long long)p.M * p.ldc) return hipErrorInvalidValue;
    if (in_launch) { p.counters = counters; p.Cf = p.C; p.cf_batch_stride = p.c_batch_stride; }
    else if (p.epi.sq_partials) { p.epi.sq_stride = red_blocks; r.epi.sq_stride = red_blocks; }
    p.C = slab; p.c_split_stride = r.split_stride; p.c_batch_stride = (long long)p.M * p.ldc;
  }
  hipError_t e;
  e = gemm_dispatch(st, p, akm, bkm, pl);
  if (e != hipSuccess || pl.nsplit == 1 || in_launch) return e;
  GANMF_LAUNCH(splitk_reduce_kernel, dim3(red_blocks, p.nbatch), dim3(256), 0, st, r);
  return hipGetLastError();
}

// staged bf16 kernels (gemm_bf16s.hpp, included by the translation unit after this header)
inline hipError_t gemm_dispatch_staged(hipStream_t st, const GemmP& p, bool akm, bool bkm, const GemmPlan& pl);

inline hipError_t gemm_dispatch_persist(hipStream_t st, const GemmP& p, bool akm, bool bkm, const GemmPlan& pl);

// Blocked tile order for a plain product with many tiles (tile_coords): of the four cuts xb_m x xb_n = 8 take the one whose
// eight L2s fetch the fewest operand bytes, with the rectangle walked in bands whose A rows fit ~1.5 MB of a 4 MB L2; B
// panels are streamed once per band.
inline void choose_tile_order(GemmP& p, const GemmPlan& pl) {
  p.xb_m = p.xb_n = p.xb_band = 0;
  if (pl.tile_order == 0 || pl.nsplit != 1 || p.nbatch != 1) return;
  const long long tiles = (long long)pl.tiles_m * pl.tiles_n;
  if (pl.tile_order == 1 && tiles < 3LL * GEMM_CUS) return;
  const double tile_bytes = 4.0 * pl.tile * std::max(p.K, 1);
  const int band = (int)std::max(1.0, std::min(1.5e6 / tile_bytes, 1e6));
  double best = 1e300;
  for (int xm = 1; xm <= 8; xm *= 2) {
    const int xn = 8 / xm;
    double fetch = 0;
    for (int i = 0; i < xm; ++i)
      for (int j = 0; j < xn; ++j) {
        const int bm = part_begin(pl.tiles_m, xm, i + 1) - part_begin(pl.tiles_m, xm, i);
        const int bn = part_begin(pl.tiles_n, xn, j + 1) - part_begin(pl.tiles_n, xn, j);
        const int bands = (bm + band - 1) / std::max(band, 1);
        fetch += tile_bytes * (bm + (double)bands * bn);
      }
    if (fetch < best) { best = fetch; p.xb_m = xm; p.xb_n = xn; }
  }
  p.xb_band = band;
}

#ifdef GANMF_PERSIST_DIAG_BUILD
// (make DIAG=1, GANMF_GEMM_STAMPS=1) four s_memrealtime stamps per workgroup of a 16-wave GEMM launch -- entry, first K-tile landed, K loop done,
// stores drained -- and their distribution over the launch's workgroups, for the first three launches of every shape: gemm_f32_mfma<.., 4>,
// gemm_bf16k_mfma and gemm_bf16w_mfma (profiles/r03_gemm_stamps.md, profiles/r06_launch_fixed_part.md)
inline bool gemm_stamps_on() { const char* v = getenv("GANMF_GEMM_STAMPS"); return v && atoi(v) != 0; }
inline bool gemm_stamps_begin(GemmP& p, int grid, hipStream_t st) {
  static unsigned long long* dbg = nullptr;
  static int cap = 0;
  if (grid > cap) { if (dbg) (void)hipFree(dbg); if (hipMalloc((void**)&dbg, (size_t)grid * 32) != hipSuccess) return false; cap = grid; }
  (void)hipMemsetAsync(dbg, 0, (size_t)grid * 32, st);
  p.stamps = dbg;
  return true;
}
inline void gemm_stamps_report(const GemmP& p, int grid, hipStream_t st, bool akm, bool bkm, const char* kernel) {
  (void)hipStreamSynchronize(st);
  static std::vector<long long> seen;
  const long long key = ((long long)p.M << 40) ^ ((long long)p.N << 20) ^ p.K ^ ((long long)akm << 62) ^ ((long long)bkm << 61) ^ ((long long)p.epi.kind << 56);
  if (std::count(seen.begin(), seen.end(), key) >= 3) return;
  seen.push_back(key);
  std::vector<unsigned long long> hs((size_t)grid * 4);
  (void)hipMemcpy(hs.data(), p.stamps, hs.size() * 8, hipMemcpyDeviceToHost);
  unsigned long long t_min = ~0ull, t_max = 0;
  for (int b = 0; b < grid; ++b) { t_min = std::min(t_min, hs[4 * b]); t_max = std::max(t_max, hs[4 * b + 3]); }
  auto stat = [&](auto f, const char* name) {
    std::vector<double> v(grid);
    for (int b = 0; b < grid; ++b) v[b] = f(b) * 0.01;      // 100 MHz ticks -> us
    std::sort(v.begin(), v.end());
    fprintf(stderr, "  %-34s min %6.2f  median %6.2f  p90 %6.2f  max %6.2f us\n", name, v[0], v[grid / 2], v[(size_t)(grid * 0.9)], v[grid - 1]);
  };
  fprintf(stderr, "[gemm stamps] %s M=%d N=%d K=%d nsplit %d epi %d akm %d bkm %d: %d workgroups, first entry -> last exit %.2f us\n", kernel, p.M, p.N, p.K, p.nsplit,
          (int)p.epi.kind, (int)akm, (int)bkm, grid, (t_max - t_min) * 0.01);
  stat([&](int b) { return (double)(hs[4 * b] - t_min); }, "entry after the first entry");
  stat([&](int b) { return (double)(hs[4 * b + 1] - hs[4 * b]); }, "entry -> first K-tile in LDS");
  stat([&](int b) { return (double)(hs[4 * b + 2] - hs[4 * b + 1]); }, "K loop");
  stat([&](int b) { return (double)(hs[4 * b + 3] - hs[4 * b + 2]); }, "epilogue + store drain");
  stat([&](int b) { return (double)(t_max - hs[4 * b + 3]); }, "exit before the last exit");
}
#endif

inline hipError_t gemm_dispatch_skinny(hipStream_t st, const GemmP& p0, bool bkm, int rt);
inline hipError_t gemm_dispatch_skinny_n(hipStream_t st, const GemmP& p0, bool bkm);
inline hipError_t gemm_dispatch_bf16w(hipStream_t st, const GemmP& p, bool bkm, int mode);

inline hipError_t gemm_dispatch(hipStream_t st, const GemmP& p0, bool akm, bool bkm, const GemmPlan& pl) {
  if (p0.epi.kind == EPI_ADAM && !(akm && bkm)) return hipErrorInvalidValue;      // (the fused Adam row pass exists in the TN kernels only)
  if (pl.skinny) return gemm_dispatch_skinny(st, p0, bkm, pl.skinny);
  if (pl.skinny_n) return gemm_dispatch_skinny_n(st, p0, bkm);
  if (pl.wide32) return gemm_dispatch_bf16w(st, p0, bkm, pl.mode);
  if (pl.persist) return gemm_dispatch_persist(st, p0, akm, bkm, pl);
  GemmP p = p0;
  choose_tile_order(p, pl);
  if (pl.mode == MFMA_BF16 || pl.mode == MFMA_BF16X3 || pl.mode == MFMA_F16) return gemm_dispatch_staged(st, p, akm, bkm, pl);
  if (pl.tile == 128) return pl.ring >= 3 ? gemm_launch_t<128, 128, 32, 3>(st, p, akm, bkm) : gemm_launch_t<128, 128, 32, 2>(st, p, akm, bkm);
#ifdef GANMF_PERSIST_DIAG_BUILD      // two K groups: between the planner's two points (one group on a shared CU, four on a CU of its own); sweeps only
  if (pl.kg == 2) return pl.ring == 3 ? gemm_launch_t<64, 64, 64, 3, 2>(st, p, akm, bkm) : gemm_launch_t<64, 64, 64, 2, 2>(st, p, akm, bkm);
#endif
#ifdef GANMF_PERSIST_DIAG_BUILD
  // diagnostic build (make DIAG=1), GANMF_GEMM_STAMPS=1: where the time of a 16-wave launch goes -- dispatch ramp, first K-tile,
  // K loop, epilogue + store drain -- from four 100 MHz stamps per workgroup, printed for the first launches of every shape
  if (pl.kg == 4 && gemm_stamps_on()) {
    const int grid = p.tiles_m * p.tiles_n * p.nsplit * p.nbatch;
    if (!gemm_stamps_begin(p, grid, st)) return hipErrorOutOfMemory;
    const hipError_t e = gemm_launch_t<64, 64, 64, 3, 4>(st, p, akm, bkm);
    gemm_stamps_report(p, grid, st, akm, bkm, "fp32 ring, 16 waves");
    return e;
  }
#endif
  if (pl.kg == 4) return gemm_launch_t<64, 64, 64, 3, 4>(st, p, akm, bkm);
#ifdef GANMF_PERSIST_DIAG_BUILD      // a 4-slot ring and a 3-slot ring with one K group: measured level, sweeps only
  if (pl.ring == 4) return gemm_launch_t<64, 64, 64, 4>(st, p, akm, bkm);   // 128 KiB ring: three K-tiles (96 KiB) in flight
  if (pl.ring == 3) return gemm_launch_t<64, 64, 64, 3>(st, p, akm, bkm);
#endif
  return gemm_launch_t<64, 64, 64, 2>(st, p, akm, bkm);
}

}  // namespace ganmf
