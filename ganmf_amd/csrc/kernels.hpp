// HBM-bound kernels of the GANMF step: CSR row expansion, row gather, column sums, the loss /
// hinge scalar kernels and the two TF-formula Adam updates.
// All are coalesced float4 streams with wavefront (64-lane) shuffle reductions; reductions that
// feed results go through per-block partials summed in a fixed order (bitwise reproducible).
#pragma once
#include <hip/hip_runtime.h>

#include "gemm_f32.hpp"

namespace ganmf {

// ---- device scalar block (floats) ----------------------------------------------------------
enum Scal : int {
  S_B1P_D = 0, S_B2P_D = 1, S_B1P_G = 2, S_B2P_G = 3,  // Adam beta powers (AdamOptimizer._finish)
  S_ALPHA_D = 4, S_ALPHA_G = 5,                        // lr_t of the step in flight
  S_ALPHA_D_ALT = 6,                                   // lr_t of odd discriminator steps (data-parallel: step i's encoder update still
                                                       // reads its lr_t on the side lane while step i+1 opens on the main lane)
  S_SUM_REAL = 8, S_SUM_FAKE = 9, S_SUM_FM = 10,        // sum of squares (local, then all-reduced)
  S_B1P_G_SAVED = 12, S_B2P_G_SAVED = 13,              // the generator's beta powers as they were before a staged discriminator pass opened
                                                       // the call's first generator pass early (rolled back when a D step fails)
  S_COUNT = 16,
  S_TAB = 16                                           // from here: lr_t of every discriminator step of a staged pass (open_steps_kernel), one slot per step
};


__device__ inline float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// block-wide sum for 256-thread blocks; result valid in thread 0
__device__ inline float block_sum_256(float v, float* red4) {
  v = wave_sum(v);
  if ((threadIdx.x & 63) == 0) red4[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red4[0] + red4[1]) + (red4[2] + red4[3]);
}

// Step prologue + CSR row expansion (replaces URM_train[uids].toarray(), GANMF.py:183-184) +
// embedding lookup (GANMF.py:82).  Block b:
//   * zero-fills row b of X (pad columns included), scatters the stored values of CSR row rows[b]
//     and sets the bias-folding ones column X[b, N] = 1 and F[b, N] = 1 (F = rows nb.. of XF);
//   * copies row rows[b] of user_embeddings into Ub[b, :].
// Block 0 / thread 0 also opens the optimizer step: lr_t from the beta powers (TF ApplyAdam),
// then advances the powers.
__device__ __forceinline__ void open_steps_body(float* __restrict__ scal, int which, int first_slot, int count, float lr);

struct DensP {
  const long long* indptr;
  const int* indices;
  const float* data;
  const int* rows;
  int nb, ncols;
  float* X;
  int ldx;
  const float* Uemb;
  int ldk;
  float* Ub;
  float* scal;
  int which, alpha_idx;
  float lr;
  int uid_col, row_offset;
  int nseg;                      // > 1 (stand-alone launch on wide rows): nseg workgroups per row, one column segment each
  int stage_b;                   // > 0: the rows of a whole pass (stage_pass): list row b belongs to minibatch b / stage_b and lands in that
                                 // minibatch's [real ; generated] block of 2 * stage_b rows; no embedding copy, no optimizer step opened
  // staged pass only: block 0 also writes lr_t of the pass's discriminator steps (open_d_count slots from open_d_slot) and, when the call's
  // first generator pass runs its all-rows Adam per pass, of that pass's steps too -- two one-thread launches less per call
  int open_d_slot, open_d_count, open_g_slot, open_g_count;
  float lr_d, lr_g;
};

// (block b of the row expansion, any block size: also runs as extra workgroups of the generator GEMM's launch, gemm_multi.hpp)
__device__ __forceinline__ void densify_row_body(const DensP& d, const int bid) {
  const int nseg = d.nseg > 1 ? d.nseg : 1;
  const int b = bid / nseg, seg = bid % nseg;
  const bool staged = d.stage_b > 0;
  if (staged && bid == 0 && threadIdx.x == 0) {
    if (d.open_d_count > 0) open_steps_body(d.scal, 0, d.open_d_slot, d.open_d_count, d.lr_d);
    if (d.open_g_count > 0) {
      d.scal[S_B1P_G_SAVED] = d.scal[S_B1P_G]; d.scal[S_B2P_G_SAVED] = d.scal[S_B2P_G];
      open_steps_body(d.scal, 1, d.open_g_slot, d.open_g_count, d.lr_g);
    }
  }
  if (!staged && d.alpha_idx >= 0 && bid == 0 && threadIdx.x == 0) {      // (alpha_idx < 0: the pass opened its steps at once, open_steps_kernel)
    const int o = d.which ? S_B1P_G : S_B1P_D;
    const float b1p = d.scal[o], b2p = d.scal[o + 1];
    d.scal[d.alpha_idx] = d.lr * sqrtf(1.f - b2p) / (1.f - b1p);
    d.scal[o] = b1p * ADAM_B1;
    d.scal[o + 1] = b2p * ADAM_B2;
  }
  const int r = d.rows[b];
  const size_t xrow = staged ? (size_t)b + (size_t)(b / d.stage_b) * d.stage_b : (size_t)b;      // row of X that receives list row b
  const size_t frow = staged ? xrow + d.stage_b : (size_t)d.nb + b;                                // ... and its generated twin
  // this workgroup's columns [c0, c1) of the row (a multiple of four floats; the whole padded row when nseg == 1)
  const int segw = ((d.ldx / 4 + nseg - 1) / nseg) * 4;
  const int c0 = seg * segw, c1 = min(d.ldx, c0 + segw);
  float* x = d.X + xrow * d.ldx;
  float4* xr = reinterpret_cast<float4*>(x + c0);
  for (int c = threadIdx.x; c < (c1 - c0) / 4; c += blockDim.x) xr[c] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (seg == 0 && d.Ub) {
    const float4* us = reinterpret_cast<const float4*>(d.Uemb + (size_t)r * d.ldk);
    float4* ud = reinterpret_cast<float4*>(d.Ub + (size_t)b * d.ldk);
    for (int c = threadIdx.x; c < d.ldk / 4; c += blockDim.x) ud[c] = us[c];
  }
  __syncthreads();
  const long long s = d.indptr[r], e = d.indptr[r + 1];
  for (long long j = s + threadIdx.x; j < e; j += blockDim.x) {
    const int col = d.indices[j];
    if (col >= c0 && col < c1) x[col] = d.data[j];
  }
  if (threadIdx.x == 0) {
    if (d.ncols >= c0 && d.ncols < c1) x[d.ncols] = 1.0f;
    if (seg == 0) d.X[frow * d.ldx + d.ncols] = 1.0f;
    if (d.uid_col >= 0) {   // DisGANMF conditions D on float(uid) (DisGANMF.py:59,110-111)
      if (d.uid_col >= c0 && d.uid_col < c1) x[d.uid_col] = (float)(d.row_offset + r);
      if (seg == 0) d.X[frow * d.ldx + d.uid_col] = (float)(d.row_offset + r);
    }
  }
}

__global__ __launch_bounds__(256) void densify_rows_kernel(const DensP d) { densify_row_body(d, (int)blockIdx.x); }

// SURVEY 8(f)-3, sparse-aware real path of the GENERATOR step.  In a G update the real rows X are needed for ONE thing:
// their encodings Er = X.We + be in the feature-matching term (GANMF.py:134); nothing else reads X.  For a sparse
// binary-ish URM that is a CSR row-sum, so the [B, N] densify of X and the real half of the encode GEMM both go away.
// Block b = batch row b:  Er[b, :] = be + sum_j data[j] * We[idx[j], :]  (CSR order inside a group, groups summed in
// index order: deterministic), the embedding gather Ub[b, :] = U[rows[b], :], the ones column of the generated row
// (F = rows nb.. of XF) and -- block 0 -- the opening of the optimizer step, i.e. everything densify_rows_kernel does
// except writing X.  256 threads = G groups x C4 float4 columns; group g takes the stored entries j = g, g + G, ...
__global__ __launch_bounds__(256) void sparse_front_kernel(const long long* __restrict__ indptr,
                                                           const int* __restrict__ indices,
                                                           const float* __restrict__ data,
                                                           const int* __restrict__ rows, int nb, int ncols,
                                                           float* __restrict__ XF, int ldx,
                                                           const float* __restrict__ Uemb, int ldk,
                                                           float* __restrict__ Ub, float* __restrict__ scal, int which,
                                                           int alpha_idx, float lr, const float* __restrict__ We, int lde,
                                                           int e, float* __restrict__ E) {
  __shared__ float4 part[256];
  const int b = blockIdx.x, tid = threadIdx.x;
  if (alpha_idx >= 0 && b == 0 && tid == 0) {
    const int o = which ? S_B1P_G : S_B1P_D;
    const float b1p = scal[o], b2p = scal[o + 1];
    scal[alpha_idx] = lr * sqrtf(1.f - b2p) / (1.f - b1p);
    scal[o] = b1p * ADAM_B1;
    scal[o + 1] = b2p * ADAM_B2;
  }
  const int r = rows[b];
  const float4* us = reinterpret_cast<const float4*>(Uemb + (size_t)r * ldk);
  if (Ub) {      // (nullptr: the embeddings of the pass were staged in schedule order, adam_rows_advance_kernel)
    float4* ud = reinterpret_cast<float4*>(Ub + (size_t)b * ldk);
    for (int c = tid; c < ldk / 4; c += 256) ud[c] = us[c];
  }
  if (tid == 0) XF[(size_t)(nb + b) * ldx + ncols] = 1.0f;
  const long long s = indptr[r], en = indptr[r + 1];
  const int c4n = (e + 3) / 4;                       // float4 columns that hold encodings (We rows are zero-padded to lde)
  const float* bias = We + (size_t)ncols * lde;      // encoder bias = row N of We_ext
  for (int c0 = 0; c0 < c4n; c0 += 256) {
    const int width = min(256, c4n - c0);
    const int G = 256 / width, g = tid / width, c = c0 + tid % width;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (g < G) {
#pragma unroll 8
      for (long long j = s + g; j < en; j += G) {      // (index, value and row fetches of eight entries in flight)
        const float w = data[j];
        const float4 a = *reinterpret_cast<const float4*>(We + (size_t)indices[j] * lde + 4 * c);
        acc.x += w * a.x; acc.y += w * a.y; acc.z += w * a.z; acc.w += w * a.w;
      }
    }
    part[tid] = acc;
    __syncthreads();
    if (tid < width) {
      float4 t = *reinterpret_cast<const float4*>(bias + 4 * c);
      for (int q = 0; q < G; ++q) {
        const float4 a = part[q * width + tid];
        t.x += a.x; t.y += a.y; t.z += a.z; t.w += a.w;
      }
      float* dst = E + (size_t)b * lde + 4 * c;     // never the ones column E[:, e]
      if (4 * c + 3 < e) *reinterpret_cast<float4*>(dst) = t;
      else {
        dst[0] = t.x;
        if (4 * c + 1 < e) dst[1] = t.y;
        if (4 * c + 2 < e) dst[2] = t.z;
      }
    }
    __syncthreads();
  }
}

// SURVEY 8(f)-3, sparse-aware real path of the DISCRIMINATOR step: the real rows' share of the encoder gradient,
//     S[j, :] = sum_{b < nb} X[b, j] * dE_r[b, :]          (gWe = X^T.dE_r + F^T.dE_f, GANMF.py:138; X = the batch's real rows)
// from the CSC form of the WHOLE matrix (column j = the rows u that store item j, ascending): u is in this batch iff
// start <= pos[u] < start + nb (pos = inverse of the epoch permutation), its batch row is pos[u] - start.  One workgroup per
// column, 256 threads = G entry groups x W float4 columns; group g takes the column's entries g, g + G, ... in order and the
// G partial sums meet in LDS in group order: no float atomics, the same sum on every run and every rank.  Columns without
// batch rows get zeros (S needs no memset).  Workgroups ncols .. ncols + CSC_BIAS_PARTS - 1 write partial column sums of dE_r
// (rows b = part, part + CSC_BIAS_PARTS, ...): the real rows' share of the encoder-BIAS gradient, added up by the consumer
// (gemm_f32.hpp sparse_rows_quad).  The consumer is the epilogue of the encoder-gradient GEMM, which then runs over the
// generated rows only.
constexpr int CSC_BIAS_PARTS = 16;
__global__ __launch_bounds__(256) void csc_rows_kernel(const long long* __restrict__ colptr, const int* __restrict__ rowidx,
                                                       const float* __restrict__ val, const int* __restrict__ pos, int start,
                                                       int nb, int ncols, const float* __restrict__ dEr, int ld, int e,
                                                       float* __restrict__ S) {
  __shared__ float4 part[256];
  const int j = blockIdx.x, tid = threadIdx.x;
  const int c4n = (e + 3) / 4;
  const bool bias = j >= ncols;
  const long long a = bias ? 0 : colptr[j], z = bias ? 0 : colptr[j + 1];
  for (int c0 = 0; c0 < c4n; c0 += 256) {
    const int width = min(256, c4n - c0);
    const int G = 256 / width, g = tid / width, c = c0 + tid % width;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (g < G) {
      if (!bias) {
#pragma unroll 4
        for (long long i = a + g; i < z; i += G) {      // (branch-free: four entries' lookups and row fetches in flight)
          const int b = pos[rowidx[i]] - start;
          const bool in = b >= 0 && b < nb;
          const float w = in ? val[i] : 0.f;
          const float4 d = *reinterpret_cast<const float4*>(dEr + (size_t)(in ? b : 0) * ld + 4 * c);
          acc.x += w * d.x; acc.y += w * d.y; acc.z += w * d.z; acc.w += w * d.w;
        }
      } else {
#pragma unroll 4
        for (int b = (j - ncols) + g * CSC_BIAS_PARTS; b < nb; b += G * CSC_BIAS_PARTS) {
          const float4 d = *reinterpret_cast<const float4*>(dEr + (size_t)b * ld + 4 * c);
          acc.x += d.x; acc.y += d.y; acc.z += d.z; acc.w += d.w;
        }
      }
    }
    part[tid] = acc;
    __syncthreads();
    if (tid < width) {
      float4 t = part[tid];
      for (int q = 1; q < G; ++q) {
        const float4 u = part[q * width + tid];
        t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
      }
      *reinterpret_cast<float4*>(S + (size_t)j * ld + 4 * c) = t;      // (pad columns of a row: sums of zero pads = 0)
    }
    __syncthreads();
  }
}

// Opens an optimizer step without any rows (a data-parallel rank that ran out of rows).
__global__ void open_step_kernel(float* __restrict__ scal, int which, int alpha_idx, float lr) {
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    const int o = which ? S_B1P_G : S_B1P_D;
    const float b1p = scal[o], b2p = scal[o + 1];
    scal[alpha_idx] = lr * sqrtf(1.f - b2p) / (1.f - b1p);
    scal[o] = b1p * ADAM_B1;
    scal[o + 1] = b2p * ADAM_B2;
  }
}

// Opens `count` consecutive optimizer steps at once: lr_t of step i -> scal[first_slot + i], the beta powers advanced `count` times
// -- the same float operations, in the same order, as `count` single openings (staged discriminator pass: no step of it has a
// row-expansion launch to open it).
__device__ __forceinline__ void open_steps_body(float* __restrict__ scal, int which, int first_slot, int count, float lr) {
  const int o = which ? S_B1P_G : S_B1P_D;
  float b1p = scal[o], b2p = scal[o + 1];
  for (int i = 0; i < count; ++i) {
    scal[first_slot + i] = lr * sqrtf(1.f - b2p) / (1.f - b1p);
    b1p = b1p * ADAM_B1;
    b2p = b2p * ADAM_B2;
  }
  scal[o] = b1p;
  scal[o + 1] = b2p;
}
// a failed discriminator pass: the generator steps that its staged row expansion opened ahead of time never ran
__global__ void restore_g_powers_kernel(float* __restrict__ scal) {
  if (blockIdx.x == 0 && threadIdx.x == 0) { scal[S_B1P_G] = scal[S_B1P_G_SAVED]; scal[S_B2P_G] = scal[S_B2P_G_SAVED]; }
}
__global__ void open_steps_kernel(float* __restrict__ scal, int which, int first_slot, int count, float lr) {
  if (blockIdx.x == 0 && threadIdx.x == 0) open_steps_body(scal, which, first_slot, count, lr);
}

// dst[b, :] = src[rows[b], :]   (embedding_lookup, GANMF.py:82)
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ src, int ld,
                                                          const int* __restrict__ rows, int nrows,
                                                          float* __restrict__ dst) {
  const int c4 = ld / 4;
  const long long total = (long long)nrows * c4;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int b = (int)(i / c4), c = (int)(i % c4);
    reinterpret_cast<float4*>(dst + (size_t)b * ld)[c] =
        reinterpret_cast<const float4*>(src + (size_t)rows[b] * ld)[c];
  }
}

// Several reproducible sums in ONE launch:
//   dst[e.dst] (+)= sum_i e.p[i], i < e.n     -- accumulate=0 overwrites, 1 adds
// One block per destination (blockIdx.x = dst; launch max dst + 1 blocks): a block walks the entries of ITS destination in
// entry order (the accumulate chain of a destination keeps its order), destinations run side by side.
struct MultiRedEntry { const float* p; int n; int dst; int accumulate; };
constexpr int MULTIRED_MAX = 8;
struct MultiRed { MultiRedEntry e[MULTIRED_MAX]; int count; float* out; };

__global__ __launch_bounds__(256) void multi_reduce_kernel(const MultiRed mr) {
  __shared__ float red[4];
  for (int z = 0; z < mr.count; ++z) {
    const MultiRedEntry e = mr.e[z];
    if (e.dst != (int)blockIdx.x) continue;      // uniform over the block
    float s = 0.f;
    for (int i = threadIdx.x; i < e.n; i += 256) s += e.p[i];
    const float t = block_sum_256(s, red);
    if (threadIdx.x == 0) mr.out[e.dst] = e.accumulate ? mr.out[e.dst] + t : t;
    __syncthreads();
  }
}

// Per-step loss parts, once per epoch: block i sums the four partial segments (cap floats each, unused
// entries zero) that step i's kernels left in its arena slot.  mode 0 (D): parts[i][2] = seg2 + seg3
// (sum theta_D^2 of the two folded tensors); mode 1 (G): parts[i][s] = seg_s, s = 0..3.
__global__ __launch_bounds__(256) void finish_parts_kernel(const float* __restrict__ arena, int cap, int mode,
                                                           float* __restrict__ parts) {
  __shared__ float red[4];
  const float* a = arena + (size_t)blockIdx.x * 4 * cap;
  float tot[4];
  for (int sgm = 0; sgm < 4; ++sgm) {
    float s = 0.f;
    for (int i = threadIdx.x; i < cap; i += 256) s += a[(size_t)sgm * cap + i];
    tot[sgm] = block_sum_256(s, red);
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    float* p = parts + (size_t)blockIdx.x * 4;
    if (mode == 0) p[2] = tot[2] + tot[3];
    else { p[0] = tot[0]; p[1] = tot[1]; p[2] = tot[2]; p[3] = tot[3]; }
  }
}

// DisGANMF: block = step; its arena slot holds nseg segments of cap floats (zero where nothing was written):
//   discriminator step (mode 0): seg 0 / 1 = per-row cross-entropies of the real / generated rows, seg 4 + t = sum(theta^2)
//                                partials of tensor t  ->  parts = {sum seg 0, sum seg 1, sum over the tensors in order}
//   generator step (mode 1):     seg 0 = cross-entropies of the generated rows, seg 1 = feature-matching partials,
//                                seg 2 / 3 = sum(U^2) / sum(V^2) partials  ->  parts = the four segment sums
__global__ __launch_bounds__(256) void finish_dis_parts_kernel(const float* __restrict__ arena, long long stride, int cap,
                                                               int nseg, int mode, float* __restrict__ parts) {
  __shared__ float red[4];
  const float* a = arena + (size_t)blockIdx.x * stride;
  float out[4] = {0.f, 0.f, 0.f, 0.f};
  for (int sgm = 0; sgm < nseg; ++sgm) {
    if (mode == 0 && (sgm == 2 || sgm == 3)) continue;
    if (mode == 1 && sgm >= 4) break;
    float s = 0.f;
    for (int i = threadIdx.x; i < cap; i += 256) s += a[(size_t)sgm * cap + i];
    const float t = block_sum_256(s, red);
    __syncthreads();
    if (mode == 1 || sgm < 2) out[sgm] = t;
    else out[2] += t;
  }
  if (threadIdx.x == 0) {
    float* p = parts + (size_t)blockIdx.x * 4;
    p[0] = out[0]; p[1] = out[1]; p[2] = out[2];
    if (mode == 1) p[3] = out[3];
  }
}

// column `col` of a [n][4] parts array <-> a contiguous vector (data-parallel: only the partial-sum columns are all-reduced)
__global__ void col_copy_kernel(float* __restrict__ parts, float* __restrict__ vec, long long n, int col, int to_vec) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    if (to_vec) vec[i] = parts[4 * i + col];
    else parts[4 * i + col] = vec[i];
  }
}

// Discriminator scalars (GANMF.py:131-132): Lr, Lf, hinge and the backward row scales
// rs[r] = c_path * 2/(B*N) with c_real = 1 + m*[h>0], c_fake = -[h>0]; loss_parts[0] = Lr + max(0,h).
// Every block recomputes the two sums from the partials in the same fixed order (identical result
// in every block) and then scales its slice of the encodings: Es = rs (.) E, the A operand of the
// decoder-gradient GEMM (its ones column becomes rs, which yields the decoder-bias gradient).
// presummed = 1: scal[S_SUM_REAL/FAKE] already hold the all-reduced sums (data-parallel).
struct DCoefP {
  float* scal;
  const float* partials;
  int np, pstride, presummed;
  float m;
  int b_local;
  float inv_bn;               // 1 / (B_global * N)
  const float* E;
  float* Es;
  int lde;
  float* rs;
  float* loss_parts;
};

// (block bx of nbx; any block size that is a multiple of 256: the two sums are formed by the first 256 threads in the order of
// the 256-thread kernel, so that the launch that carries this body as extra blocks -- gemm_multi.hpp, dE GEMM + d_coef --
// reproduces it bit for bit; `red` = 6 floats of LDS)
__device__ __forceinline__ void d_coef_body(const DCoefP& p, const int bx, const int nbx, float* __restrict__ red) {
  float* sums = red + 4;
  if (!p.presummed) {
    for (int z = 0; z < 2; ++z) {
      float s = 0.f;
      if (threadIdx.x < 256)
        for (int i = threadIdx.x; i < p.np; i += 256) s += p.partials[(size_t)z * p.pstride + i];
      s = wave_sum(s);
      if (threadIdx.x < 256 && (threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
      __syncthreads();
      if (threadIdx.x == 0) sums[z] = (red[0] + red[1]) + (red[2] + red[3]);
      __syncthreads();
    }
  } else {
    if (threadIdx.x == 0) { sums[0] = p.scal[S_SUM_REAL]; sums[1] = p.scal[S_SUM_FAKE]; }
    __syncthreads();
  }
  const float Lr = sums[0] * p.inv_bn, Lf = sums[1] * p.inv_bn;
  const float h = p.m * Lr - Lf;
  const bool on = h > 0.f;
  const float cr = (on ? 1.f + p.m : 1.f) * (2.f * p.inv_bn);
  const float cf = (on ? -1.f : 0.f) * (2.f * p.inv_bn);
  const int c4 = p.lde / 4;
  const long long total = (long long)2 * p.b_local * c4;
  for (long long i = (long long)bx * blockDim.x + threadIdx.x; i < total; i += (long long)nbx * blockDim.x) {
    const int r = (int)(i / c4);
    const float sc = r < p.b_local ? cr : cf;
    float4 v = reinterpret_cast<const float4*>(p.E)[i];
    v.x *= sc; v.y *= sc; v.z *= sc; v.w *= sc;
    reinterpret_cast<float4*>(p.Es)[i] = v;
  }
  if (bx == 0) {
    for (int r = threadIdx.x; r < 2 * p.b_local; r += blockDim.x) p.rs[r] = r < p.b_local ? cr : cf;
    if (threadIdx.x == 0) {
      p.loss_parts[0] = Lr + fmaxf(0.f, h);
      p.loss_parts[1] = on ? 1.f : 0.f;
    }
  }
}

__global__ __launch_bounds__(256) void d_coef_kernel(const DCoefP p) {
  __shared__ float red[6];
  d_coef_body(p, (int)blockIdx.x, (int)gridDim.x, red);
}

// ApplyAdam (dense):  g' = g + reg*theta ; m += (g'-m)(1-b1) ; v += (g'^2-v)(1-b2) ;
// theta -= (m*alpha)/(sqrt(v)+eps).  Optionally accumulates sum(theta_old^2) per block.
__global__ __launch_bounds__(256) void adam_dense_kernel(float* __restrict__ th, float* __restrict__ mo,
                                                         float* __restrict__ vo, const float* __restrict__ g,
                                                         long long n4, const float* __restrict__ scal,
                                                         int alpha_idx, float reg, float* __restrict__ sq_partials) {
  __shared__ float red[4];
  const float alpha = scal[alpha_idx];
  float sq = 0.f;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4;
       i += (long long)gridDim.x * blockDim.x) {
    float4 t = reinterpret_cast<float4*>(th)[i];
    float4 m = reinterpret_cast<float4*>(mo)[i];
    float4 v = reinterpret_cast<float4*>(vo)[i];
    const float4 gg = reinterpret_cast<const float4*>(g)[i];
    float* tp = &t.x; float* mp = &m.x; float* vp = &v.x; const float* gp = &gg.x;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float x = tp[j];
      sq += x * x;
      const float gr = gp[j] + reg * x;
      mp[j] += (gr - mp[j]) * (1.f - ADAM_B1);
      vp[j] += (gr * gr - vp[j]) * (1.f - ADAM_B2);
      tp[j] = adam_step(x, mp[j] * alpha, vp[j]);
    }
    reinterpret_cast<float4*>(th)[i] = t;
    reinterpret_cast<float4*>(mo)[i] = m;
    reinterpret_cast<float4*>(vo)[i] = v;
  }
  if (sq_partials) {
    const float s = block_sum_256(sq, red);
    if (threadIdx.x == 0) sq_partials[blockIdx.x] = s;
  }
}

// one element of the all-rows update below (shared with the per-pass forms, adam_rows_advance_kernel / adam_rows_flush_kernel)
__device__ __forceinline__ void adam_row_elem(float g, float reg, float alpha, float& th, float& m, float& v, float& sq) {
  const float x = th;
  sq += x * x;
  const float gr = g + reg * x;
  m = m * ADAM_B1 + gr * (1.f - ADAM_B1);
  v = v * ADAM_B2 + (gr * gr) * (1.f - ADAM_B2);
  th = adam_step(x, alpha * m, v);
}

// AdamOptimizer._apply_sparse_shared over ALL rows of user_embeddings (TF-1 Adam is not lazy,
// SURVEY Appendix B.5): rows of the current batch take their gradient from gUb, others g = 0.
//   m = m*b1 + g'(1-b1) ; v = v*b2 + g'^2(1-b2) ; theta -= alpha*m/(sqrt(v)+eps)
// pos[r] = position of row r in this epoch's permutation (-1 if absent); batch = [start, start+nb).
// The batch gradient arrives as gsplit split-K slabs of the gUb GEMM (slab s at gb + s * gstride), summed here in split
// order -- exactly what splitk_reduce_kernel would have written, without its launch (only splits that kernel sums one thread
// per element, reduce_groups == 1: a deep split behind this small output keeps its reduce launch).
__global__ __launch_bounds__(256) void adam_rows_kernel(float* __restrict__ th, float* __restrict__ mo,
                                                        float* __restrict__ vo, const float* __restrict__ gb,
                                                        int gsplit, long long gstride,
                                                        const int* __restrict__ pos, int start, int nb,
                                                        int nrows, int ld, const float* __restrict__ scal,
                                                        int alpha_idx, float reg, float* __restrict__ sq_partials) {
  __shared__ float red[4];
  const float alpha = scal[alpha_idx];
  const int c4 = ld / 4;
  const long long total = (long long)nrows * c4;
  float sq = 0.f;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int r = (int)(i / c4), c = (int)(i % c4);
    const int slot = pos[r] - start;
    float4 gg = make_float4(0.f, 0.f, 0.f, 0.f);
    if (slot >= 0 && slot < nb) {
      const float* gp0 = gb + (size_t)slot * ld + 4 * c;
      gg = *reinterpret_cast<const float4*>(gp0);
#pragma unroll 8
      for (int sp = 1; sp < gsplit; ++sp) {   // independent loads: several slabs in flight
        const float4 q = *reinterpret_cast<const float4*>(gp0 + (size_t)sp * gstride);
        gg.x += q.x; gg.y += q.y; gg.z += q.z; gg.w += q.w;
      }
    }
    float4 t = reinterpret_cast<float4*>(th)[i];
    float4 m = reinterpret_cast<float4*>(mo)[i];
    float4 v = reinterpret_cast<float4*>(vo)[i];
    float* tp = &t.x; float* mp = &m.x; float* vp = &v.x; const float* gp = &gg.x;
#pragma unroll
    for (int j = 0; j < 4; ++j) adam_row_elem(gp[j], reg, alpha, tp[j], mp[j], vp[j], sq);
    reinterpret_cast<float4*>(th)[i] = t;
    reinterpret_cast<float4*>(mo)[i] = m;
    reinterpret_cast<float4*>(vo)[i] = v;
  }
  if (sq_partials) {
    const float s = block_sum_256(sq, red);
    if (threadIdx.x == 0) sq_partials[blockIdx.x] = s;
  }
}

// ---- the all-rows update of user_embeddings, once per generator PASS instead of once per step (g_reg == 0) ---------------------
// During a generator pass a row of U is read exactly once -- by the step whose minibatch holds it -- and (with g_reg = 0) receives
// a non-zero gradient exactly once, at that step; every other step only decays its moments and moves it by lr_t * m / (sqrt(v) + eps).
// Those T updates per element depend on nothing but the element's own state, so they need not be T passes over 6 x U x k floats:
//   adam_rows_advance_kernel (front of the pass): row at schedule position a, used by step s = pos_step[a], advanced by s
//       zero-gradient steps -> Ub_sched[a] = the embedding that step will read (the parameter itself is not touched);
//   adam_rows_flush_kernel (end of the pass): every row of U through all T steps, the gradient of ITS step summed from that step's
//       gUb slabs in split order -- theta, m, v read and written once per pass.
// Same float operations in the same order per element as T launches of adam_rows_kernel (adam_row_elem): bit-identical.
struct LazyStep {
  const float* g;          // gUb of the step: nsplit slabs [nb, ld] (slab q at g + q * gstride)
  long long gstride;
  int nsplit, start, nb;   // the step's minibatch = schedule positions [start, start + nb)
};

__global__ __launch_bounds__(256) void adam_rows_advance_kernel(const float* __restrict__ th, const float* __restrict__ mo,
                                                                const float* __restrict__ vo, const int* __restrict__ perm,
                                                                const int* __restrict__ pos_step, int n, int ld,
                                                                const float* __restrict__ scal, int tab, float reg,
                                                                float* __restrict__ out) {
  const int c4 = ld / 4;
  const long long total = (long long)n * c4;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int a = (int)(i / c4), c = (int)(i % c4);
    const size_t src = (size_t)perm[a] * c4 + c;
    const int s = pos_step[a];
    float4 t = reinterpret_cast<const float4*>(th)[src];
    float4 m = reinterpret_cast<const float4*>(mo)[src];
    float4 v = reinterpret_cast<const float4*>(vo)[src];
    float* tp = &t.x; float* mp = &m.x; float* vp = &v.x;
    float sq = 0.f;
    for (int tt = 0; tt < s; ++tt) {
      const float alpha = scal[tab + tt];
#pragma unroll
      for (int j = 0; j < 4; ++j) adam_row_elem(0.f, reg, alpha, tp[j], mp[j], vp[j], sq);
    }
    reinterpret_cast<float4*>(out)[i] = t;
  }
}

__global__ __launch_bounds__(256) void adam_rows_flush_kernel(float* __restrict__ th, float* __restrict__ mo, float* __restrict__ vo,
                                                              const int* __restrict__ pos, const int* __restrict__ pos_step,
                                                              const LazyStep* __restrict__ steps, int T, int nrows, int ld,
                                                              const float* __restrict__ scal, int tab, float reg) {
  const int c4 = ld / 4;
  const long long total = (long long)nrows * c4;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int r = (int)(i / c4), c = (int)(i % c4);
    const int a = pos[r];
    const int s = a >= 0 ? pos_step[a] : -1;
    float4 gg = make_float4(0.f, 0.f, 0.f, 0.f);
    if (s >= 0) {
      const LazyStep st = steps[s];
      const float* gp0 = st.g + (size_t)(a - st.start) * ld + 4 * c;
      gg = *reinterpret_cast<const float4*>(gp0);
#pragma unroll 8
      for (int sp = 1; sp < st.nsplit; ++sp) {
        const float4 q = *reinterpret_cast<const float4*>(gp0 + (size_t)sp * st.gstride);
        gg.x += q.x; gg.y += q.y; gg.z += q.z; gg.w += q.w;
      }
    }
    float4 t = reinterpret_cast<float4*>(th)[i];
    float4 m = reinterpret_cast<float4*>(mo)[i];
    float4 v = reinterpret_cast<float4*>(vo)[i];
    float* tp = &t.x; float* mp = &m.x; float* vp = &v.x; const float* gp = &gg.x;
    float sq = 0.f;
    for (int tt = 0; tt < T; ++tt) {
      const float alpha = scal[tab + tt];
      const bool mine = tt == s;
#pragma unroll
      for (int j = 0; j < 4; ++j) adam_row_elem(mine ? gp[j] : 0.f, reg, alpha, tp[j], mp[j], vp[j], sq);
    }
    reinterpret_cast<float4*>(th)[i] = t;
    reinterpret_cast<float4*>(mo)[i] = m;
    reinterpret_cast<float4*>(vo)[i] = v;
  }
}

// ---- DisGANMF discriminator head (DisGANMF.py:63,114-117) ---------------------------------------
// One wave per row: logit = [feat | 1] . wo_ext ; sigmoid cross-entropy against the row's label
// (real rows 1, generated rows 0); dlogit = (sigmoid(logit) - label) / B_global.
// row0 / nrows select the rows (D-step: all 2B; G-step: the generated half only).
__global__ __launch_bounds__(256) void dis_head_kernel(const float* __restrict__ feat, int ld, int e1,
                                                       const float* __restrict__ wo, int row0, int nrows,
                                                       int n_real, float inv_b, float* __restrict__ dlogit,
                                                       float* __restrict__ loss_real, float* __restrict__ loss_fake) {
  const int r = row0 + blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= row0 + nrows) return;
  const int lane = threadIdx.x & 63;
  float s = 0.f;
#pragma unroll 8
  for (int j = lane; j < e1; j += 64) s += feat[(size_t)r * ld + j] * wo[j];      // (loads of eight rounds in flight)
  s = wave_sum(s);
  if (lane == 0) {
    const float z = r < n_real ? 1.f : 0.f;
    // max(x,0) - x*z + log1p(exp(-|x|))   (tf.nn.sigmoid_cross_entropy_with_logits)
    const float l = fmaxf(s, 0.f) - s * z + log1pf(expf(-fabsf(s)));
    if (r < n_real) loss_real[r] = l;            // the two label groups are summed separately, once per epoch
    else loss_fake[r - n_real] = l;              // (finish_dis_parts_kernel)
    dlogit[r] = (1.f / (1.f + expf(-s)) - z) * inv_b;
  }
}

// The head inside the slab sum of the last hidden layer (round 4): when that layer's forward GEMM is split along K, its reduce launch
// is this kernel -- one workgroup per ROW sums the row's slabs (same order as splitk_reduce_body), applies the activation epilogue,
// stores the layer output, and forms the row's logit [a | 1] . wo_ext, cross-entropy and dlogit on the way: dis_head_kernel's launch
// (5-6 us at the launch floor, twice per D + G pair) disappears.  Rows outside [row0, row0 + nrows) only get their layer output
// (generator step: the real half is needed for feature matching, not for the loss).
struct HeadP {
  const float* wo;       // [e + 1] output kernel then output bias
  int row0, nrows, n_real;
  float inv_b;
  float* dlogit;
  float* loss_real;      // may be nullptr (generator step)
  float* loss_fake;
  // generator step (pair_off > 0): the workgroup of generated row r also forms the top of the backward pass for its row --
  // dz = (dlogit . wo + fmc (a_f - a_r)) act'(a_f) and the feature-matching partial sum (a_f - a_r)^2 -> fm_partials[r - pair_off],
  // with a_r = the output of the paired real row r - pair_off -- which is row-local once the row's logit exists: dis_dz_top_kernel's
  // launch disappears from the generator step too.  (The discriminator step's top also sums feat^T . dlogit over ALL rows for the
  // output layer's gradient: it keeps its kernel.)
  int pair_off;
  float fmc;
  int act;
  float* dz;             // [.., ld] top-of-backward output (generated rows)
  float* fm_partials;
  // discriminator step (round 6, pair_off == 0): dz_rows != 0 -> the workgroup of row r also writes the row-local part of the top of the
  // backward pass, dz[r, c] = dlogit[r] . wo[c] . act'(a[r, c]); the column sums over all rows (output-layer gradient, float(uid) row) ride in
  // the gradient GEMM's launch (dis_colsum_body, gemm_multi.hpp gemm_bf16s_colsum) and dis_dz_top_kernel's launch leaves the step
  int dz_rows = 0;
};

// The column part of the top of the discriminator's backward pass (dis_dz_top_kernel without the row-local dz): for 16 columns per block,
//   g_wo[c] = sum_r feat[r, c] . dlogit[r]  (c <= e; column e = the ones column: the output bias)   -> TF-Adam on wo[c] in place
//   g_u[c]  = sum_r uid[r] . dz[r, c]       (c < e; low-precision modes: the fp32 float(uid) row of W_0_ext) -> TF-Adam on that row's column c
// 256 threads = 16 columns x 16 row groups; the group partials meet in LDS and are added in group order (fixed, reproducible).
struct ColSumP {
  const float* feat; int ld; int e;
  const float* dlogit; int nrows;
  const float* dz;
  float *th, *mo, *vo;              // output layer (e + 1 values)
  const float* scal; int alpha_idx; float reg;
  float* sq_partials;               // [blocks] sum(theta_old^2), or nullptr
  const float* uid; int ldu;        // nullptr: no float(uid) part
  float *uth, *umo, *uvo, *usq;
};
constexpr int CS_COLS = 16, CS_GROUPS = 16;
inline int dis_colsum_blocks(int e) { return (e + 1 + CS_COLS - 1) / CS_COLS; }

__device__ __forceinline__ void dis_colsum_body(const ColSumP& q, const int blk, float* __restrict__ smem) {
  float (*red)[CS_COLS] = reinterpret_cast<float (*)[CS_COLS]>(smem);
  float (*red2)[CS_COLS] = reinterpret_cast<float (*)[CS_COLS]>(smem + CS_GROUPS * CS_COLS);
  const int cl = threadIdx.x % CS_COLS, g = threadIdx.x / CS_COLS;
  const int c = blk * CS_COLS + cl;
  float acc = 0.f, uacc = 0.f;
  if (c <= q.e) {
#pragma unroll 4
    for (int r = g; r < q.nrows; r += CS_GROUPS) {
      acc += q.feat[(size_t)r * q.ld + c] * q.dlogit[r];
      if (q.uid && c < q.e) uacc += q.uid[(size_t)r * q.ldu] * q.dz[(size_t)r * q.ld + c];
    }
  }
  red[g][cl] = acc;
  red2[g][cl] = uacc;
  __syncthreads();
  float sqv = 0.f, usv = 0.f;
  if (g == 0 && c <= q.e) {
    float t = red[0][cl];
#pragma unroll
    for (int k = 1; k < CS_GROUPS; ++k) t += red[k][cl];
    const float x = q.th[c];
    sqv = x * x;
    const float gr = t + q.reg * x;
    float mm = q.mo[c], vv = q.vo[c];
    mm += (gr - mm) * (1.f - ADAM_B1);
    vv += (gr * gr - vv) * (1.f - ADAM_B2);
    q.mo[c] = mm; q.vo[c] = vv;
    q.th[c] = adam_step(x, mm * q.scal[q.alpha_idx], vv);
  }
  if (q.uid && g == 1 && c < q.e) {
    float t = red2[0][cl];
#pragma unroll
    for (int k = 1; k < CS_GROUPS; ++k) t += red2[k][cl];
    const float x = q.uth[c];
    usv = x * x;
    const float gr = t + q.reg * x;
    float mm = q.umo[c], vv = q.uvo[c];
    mm += (gr - mm) * (1.f - ADAM_B1);
    vv += (gr * gr - vv) * (1.f - ADAM_B2);
    q.umo[c] = mm; q.uvo[c] = vv;
    q.uth[c] = adam_step(x, mm * q.scal[q.alpha_idx], vv);
  }
  if (threadIdx.x < 64) {      // (wave 0 = row groups 0 .. 3: sqv lives in group 0's lanes, usv in group 1's)
    if (q.sq_partials) { sqv = wave_sum(sqv); if (threadIdx.x == 0) q.sq_partials[blk] = sqv; }
    if (q.uid && q.usq) { usv = wave_sum(usv); if (threadIdx.x == 0) q.usq[blk] = usv; }
  }
}

// one row's slab sum + epilogue; returns the row's partial dot with wo over this thread's columns
__device__ __forceinline__ float reduce_row_dot(const RedP& p, const float* __restrict__ wo, int r) {
  const int n4 = (p.N + 3) >> 2;
  const EpiD& e = p.epi;
  float dot = 0.f, sq = 0.f;
  for (int c4 = threadIdx.x; c4 < n4; c4 += 256) {
    const int c = 4 * c4;
    const size_t off = (size_t)r * p.ld + c;
    float4 s = *reinterpret_cast<const float4*>(p.part + off);
#pragma unroll 8
    for (int k = 1; k < p.nsplit; ++k) {
      const float4 q = *reinterpret_cast<const float4*>(p.part + (size_t)k * p.split_stride + off);
      s.x += q.x; s.y += q.y; s.z += q.z; s.w += q.w;
    }
    float o[4] = {s.x, s.y, s.z, s.w};
    if (c + 3 < p.N) {
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = epi_apply(e, o[j], r, c + j, p.ld, nullptr, sq);
      *reinterpret_cast<float4*>(p.out + off) = make_float4(o[0], o[1], o[2], o[3]);
      const float4 w = *reinterpret_cast<const float4*>(wo + c);
      dot += o[0] * w.x; dot += o[1] * w.y; dot += o[2] * w.z; dot += o[3] * w.w;
    } else {
      for (int j = 0; j < 4 && c + j < p.N; ++j) {
        const float v = epi_apply(e, o[j], r, c + j, p.ld, nullptr, sq);
        p.out[off + j] = v;
        dot += v * wo[c + j];
      }
    }
  }
  return dot;
}

// the finished values of four columns of row r (slab sum + epilogue), nothing stored
__device__ __forceinline__ float4 reduce_row_quad(const RedP& p, int r, int c) {
  const size_t off = (size_t)r * p.ld + c;
  float4 s = *reinterpret_cast<const float4*>(p.part + off);
#pragma unroll 8
  for (int k = 1; k < p.nsplit; ++k) {
    const float4 q = *reinterpret_cast<const float4*>(p.part + (size_t)k * p.split_stride + off);
    s.x += q.x; s.y += q.y; s.z += q.z; s.w += q.w;
  }
  float sq = 0.f;
  s.x = epi_apply(p.epi, s.x, r, c, p.ld, nullptr, sq);
  s.y = c + 1 < p.N ? epi_apply(p.epi, s.y, r, c + 1, p.ld, nullptr, sq) : 0.f;
  s.z = c + 2 < p.N ? epi_apply(p.epi, s.z, r, c + 2, p.ld, nullptr, sq) : 0.f;
  s.w = c + 3 < p.N ? epi_apply(p.epi, s.w, r, c + 3, p.ld, nullptr, sq) : 0.f;
  return s;
}

// grid = the M rows of the layer output, one workgroup each
__global__ __launch_bounds__(256) void reduce_rows_head_kernel(const RedP p, const HeadP hd) {
  __shared__ float red[8];
  const int r = (int)blockIdx.x;
  float dot = reduce_row_dot(p, hd.wo, r);
  if (r < hd.row0 || r >= hd.row0 + hd.nrows) return;      // (uniform per workgroup)
  dot = wave_sum(dot);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = dot;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float sl = ((red[0] + red[1]) + (red[2] + red[3])) + hd.wo[p.N];      // + the ones column's weight (the output bias)
    const float z = r < hd.n_real ? 1.f : 0.f;
    const float l = fmaxf(sl, 0.f) - sl * z + log1pf(expf(-fabsf(sl)));          // tf.nn.sigmoid_cross_entropy_with_logits
    if (r < hd.n_real) { if (hd.loss_real) hd.loss_real[r] = l; }
    else hd.loss_fake[r - hd.n_real] = l;
    const float dl = (1.f / (1.f + expf(-sl)) - z) * hd.inv_b;
    hd.dlogit[r] = dl;
    red[4] = dl;
  }
  if (hd.pair_off <= 0 && hd.dz_rows) {      // discriminator step: the row-local part of the top of the backward pass (HeadP::dz_rows)
    __syncthreads();      // red[4]; this thread re-reads the layer outputs it stored itself (reduce_row_dot walks the same columns)
    const float dl = red[4];
    const int n4 = (p.N + 3) >> 2;
    for (int c4 = threadIdx.x; c4 < n4; c4 += 256) {
      const int c = 4 * c4;
      const size_t off = (size_t)r * p.ld + c;
      if (c + 3 < p.N) {
        const float4 a = *reinterpret_cast<const float4*>(p.out + off);
        const float4 w = *reinterpret_cast<const float4*>(hd.wo + c);
        *reinterpret_cast<float4*>(hd.dz + off) = make_float4(dl * w.x * act_grad_out(hd.act, a.x), dl * w.y * act_grad_out(hd.act, a.y),
                                                             dl * w.z * act_grad_out(hd.act, a.z), dl * w.w * act_grad_out(hd.act, a.w));
      } else {
        for (int j = 0; j < 4 && c + j < p.N; ++j) hd.dz[off + j] = dl * hd.wo[c + j] * act_grad_out(hd.act, p.out[off + j]);
      }
    }
    return;
  }
  if (hd.pair_off <= 0) return;
  // generator step: the top of the backward pass for this (generated) row.  Its paired real row's output belongs to ANOTHER
  // workgroup of this launch: its values are formed again here from the slabs (the same sums, the same epilogue: the same bits),
  // four more slab loads per float4 instead of a dependency between workgroups.
  __syncthreads();      // red[4]; and this workgroup's stores of its own row are behind its barrier
  const float dl = red[4];
  float fm = 0.f;
  for (int c = 4 * threadIdx.x; c < p.N; c += 4 * 256) {
    const float4 ar = reduce_row_quad(p, r - hd.pair_off, c);
    const float4 af = reduce_row_quad(p, r, c);
    const float a_f[4] = {af.x, af.y, af.z, af.w}, a_r[4] = {ar.x, ar.y, ar.z, ar.w};
    float o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float d = a_f[j] - a_r[j];
      if (c + j < p.N) fm += d * d;
      o[j] = c + j < p.N ? (dl * hd.wo[c + j] + hd.fmc * d) * act_grad_out(hd.act, a_f[j]) : 0.f;
    }
    if (c + 3 < p.N) *reinterpret_cast<float4*>(hd.dz + (size_t)r * p.ld + c) = make_float4(o[0], o[1], o[2], o[3]);
    else for (int j = 0; j < 4 && c + j < p.N; ++j) hd.dz[(size_t)r * p.ld + c + j] = o[j];
  }
  fm = wave_sum(fm);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = fm;
  __syncthreads();
  if (threadIdx.x == 0) hd.fm_partials[r - hd.pair_off] = (red[0] + red[1]) + (red[2] + red[3]);
}

// Top of the backward pass, rows [row0, row0+nrows), columns [0, e1):
//   dh = dlogit[r] * wo[j] + fmc * (feat[r, j] - feat[r - pair_off, j])      (fmc != 0: G-step)
//   dz[r, j] = dh * act'(feat[r, j])                 (columns < e only)
//   gwo[j] = sum_r feat[r, j] * dlogit[r]            (D-step; column e = 1 gives the bias gradient)
//   fm partial = sum (feat_f - feat_r)^2             (G-step)
// A block takes DZ_COLS columns and splits the rows over DZ_GROUPS row groups (thread = (group, column)): 65 blocks of 1024
// threads at d_nodes = 1024 with two to four rows per thread, where round 1's 64-column blocks left 17 workgroups walking 64
// rows each (30 us of a 127 us step).  The column sums add the groups in a fixed two-level order (four consecutive groups, then the sixteen sums), the feature-matching partial the
// waves in wave order: fixed, reproducible.
constexpr int DZ_COLS = 16, DZ_GROUPS = 64;
inline int dis_dz_top_blocks(int e) { return (e + 1 + DZ_COLS - 1) / DZ_COLS; }

__global__ __launch_bounds__(DZ_COLS * DZ_GROUPS) void dis_dz_top_kernel(const float* __restrict__ feat, int ld, int e,
                                                                          const float* wo /* may alias th */,
                                                                          const float* __restrict__ dlogit, int row0, int nrows,
                                                                          int pair_off, float fmc, int act, float* __restrict__ dz,
                                                                          float* __restrict__ gwo, float* __restrict__ fm_partials,
                                                                          float* th, float* __restrict__ mo,
                                                                          float* __restrict__ vo, const float* __restrict__ scal,
                                                                          int alpha_idx, float reg, float* __restrict__ sq_partials,
                                                                          const float* __restrict__ uid = nullptr, int ldu = 0,
                                                                          float* __restrict__ uth = nullptr, float* __restrict__ umo = nullptr,
                                                                          float* __restrict__ uvo = nullptr, float* __restrict__ usq = nullptr) {
  // th != nullptr (D-step, single GPU): gwo is not stored -- TF-Adam on the output layer's column right here (this block is the
  // only reader of wo[c], and the logits were formed by the previous kernel); sum(theta_old^2) of the block -> sq_partials[block]
  // uid != nullptr (round 6; D-step of a ONE-layer discriminator in a low-precision mode, single GPU): dz written here IS dz_0, so the
  // gradient of the fp32 float(uid) row of W_0_ext, sum_m uid[m] dz[m, c] (dis_uid_grad_kernel's job), is formed in the same pass over the rows
  // and TF-Adam applied to that row's column c right here (uth / umo / uvo; nothing reads the row between the forward product and this point);
  // sum(theta_old^2) of the block -> usq[block].  dis_uid_grad_kernel's launch leaves the step.
  __shared__ float red[DZ_GROUPS][DZ_COLS], red2[DZ_GROUPS][DZ_COLS];
  static_assert(DZ_GROUPS == 64 && DZ_COLS * DZ_GROUPS >= 16 * DZ_COLS, "the two-level column sums below: 64 groups = 16 x 4");
  const int cl = threadIdx.x % DZ_COLS, g = threadIdx.x / DZ_COLS;
  const int c = blockIdx.x * DZ_COLS + cl;
  float acc = 0.f, fm = 0.f, uacc = 0.f;
  if (c <= e) {
    const float w = wo[c];
    for (int r = row0 + g; r < row0 + nrows; r += DZ_GROUPS) {
      const float a = feat[(size_t)r * ld + c];
      const float dl = dlogit[r];
      acc += a * dl;
      if (c < e) {
        float dh = dl * w;
        if (pair_off > 0) {   // G-step: feature matching against the paired real row
          const float d = a - feat[(size_t)(r - pair_off) * ld + c];
          dh += fmc * d;
          fm += d * d;
        }
        const float dzv = dh * act_grad_out(act, a);
        dz[(size_t)r * ld + c] = dzv;
        if (uid) uacc += uid[(size_t)r * ldu] * dzv;
      }
    }
  }
  // The 64 group partials of a column meet in LDS and are added in a FIXED two-level order (round 6): sixteen threads per column add four
  // consecutive groups each, then one thread adds the sixteen sums in order -- depth 4 + 16 instead of one thread walking 64 LDS reads (3 us of
  // a 5 us launch).  Both sums of the launch go through together: the output layer's gradient (thread group 0 finishes it) and, with `uid`, the
  // float(uid) row's (thread group 1).
  red[g][cl] = acc;
  if (uid) red2[g][cl] = uacc;
  __syncthreads();
  const bool lvl1 = g < 16;      // thread group g < 16 adds groups 4 g .. 4 g + 3 of its column
  float a4 = 0.f, u4 = 0.f;
  if (lvl1) {
    a4 = ((red[4 * g][cl] + red[4 * g + 1][cl]) + red[4 * g + 2][cl]) + red[4 * g + 3][cl];
    if (uid) u4 = ((red2[4 * g][cl] + red2[4 * g + 1][cl]) + red2[4 * g + 2][cl]) + red2[4 * g + 3][cl];
  }
  __syncthreads();
  if (lvl1) {
    red[g][cl] = a4;
    if (uid) red2[g][cl] = u4;
  }
  __syncthreads();
  float sqv = 0.f, usv = 0.f;
  if (g == 0 && c <= e && (gwo || th)) {
    float t = red[0][cl];
#pragma unroll
    for (int q = 1; q < 16; ++q) t += red[q][cl];
    if (th) {
      const float x = th[c];
      sqv = x * x;
      const float gr = t + reg * x;
      float mm = mo[c], vv = vo[c];
      mm += (gr - mm) * (1.f - ADAM_B1);
      vv += (gr * gr - vv) * (1.f - ADAM_B2);
      mo[c] = mm; vo[c] = vv;
      th[c] = adam_step(x, mm * scal[alpha_idx], vv);
    } else {
      gwo[c] = t;
    }
  }
  if (uid && g == 1 && c < e) {      // the float(uid) row of W_0_ext: gradient -> TF-Adam in place
    float t = red2[0][cl];
#pragma unroll
    for (int q = 1; q < 16; ++q) t += red2[q][cl];
    const float x = uth[c];
    usv = x * x;
    const float gr = t + reg * x;
    float mm = umo[c], vv = uvo[c];
    mm += (gr - mm) * (1.f - ADAM_B1);
    vv += (gr * gr - vv) * (1.f - ADAM_B2);
    umo[c] = mm; uvo[c] = vv;
    uth[c] = adam_step(x, mm * scal[alpha_idx], vv);
  }
  if (threadIdx.x < 64) {      // (wave 0 holds thread groups 0 .. 3: sqv lives in group 0's lanes, usv in group 1's)
    if (th && sq_partials) { sqv = wave_sum(sqv); if (threadIdx.x == 0) sq_partials[blockIdx.x] = sqv; }
    if (uid && usq) { usv = wave_sum(usv); if (threadIdx.x == 0) usq[blockIdx.x] = usv; }
  }
  if (fm_partials) {
    __syncthreads();
    fm = wave_sum(fm);
    float* wsum = &red[0][0];
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = fm;
    __syncthreads();
    if (threadIdx.x == 0) {
      float t = 0.f;
      for (int q = 0; q < DZ_COLS * DZ_GROUPS / 64; ++q) t += wsum[q];
      fm_partials[blockIdx.x] = t;
    }
  }
}

// DisGANMF with a low-precision K loop: the float(uid) input column stays in fp32 OUTSIDE the MFMA.  Forward:
// rank-1 term in the layer-0 epilogue (EpiD::r1_*).  Backward: this kernel, the gradient of that column's weight row,
//   out[n] = sum_m X[m, uid_col] * dz[m, n]     (rows m < rows; fixed summation order)
// One workgroup = 64 output columns x UIDG_GROUPS row groups: group q sums rows q, q + UIDG_GROUPS, ... (independent loads, all in
// flight at once), the group partials meet in LDS and are added in group order.  (One thread per column walking all rows took
// 62 us for 256 rows x 1024 columns -- a quarter of the low-precision DisGANMF step -- on four workgroups.)
// th != nullptr: the row is not stored but applied -- TF-Adam on th / mo / vo (the float(uid) row of W_0_ext, whose other rows
// were updated in the epilogue of the gradient GEMM), sum(theta_old^2) of the block's columns -> sq_partials[blockIdx.x].
constexpr int UIDG_GROUPS = 16;
__global__ __launch_bounds__(64 * UIDG_GROUPS) void dis_uid_grad_kernel(const float* __restrict__ X, int ldx, int uid_col,
                                                                        const float* __restrict__ dz, int lddz, int rows, int e,
                                                                        float* __restrict__ out, float* __restrict__ th,
                                                                        float* __restrict__ mo, float* __restrict__ vo,
                                                                        const float* __restrict__ scal, int alpha_idx, float reg,
                                                                        float* __restrict__ sq_partials) {
  __shared__ float part[UIDG_GROUPS][64];
  const int c = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int n = blockIdx.x * 64 + c;
  float s = 0.f;
  if (n < e) {
#pragma unroll 8
    for (int m = q; m < rows; m += UIDG_GROUPS) s += X[(size_t)m * ldx + uid_col] * dz[(size_t)m * lddz + n];
  }
  part[q][c] = s;
  __syncthreads();
  if (q != 0) return;
  float sq = 0.f;
  if (n < e) {
    float t = part[0][c];
#pragma unroll
    for (int g = 1; g < UIDG_GROUPS; ++g) t += part[g][c];
    if (th) {
      const float x = th[n];
      sq = x * x;
      const float gr = t + reg * x;
      float mm = mo[n], vv = vo[n];
      mm += (gr - mm) * (1.f - ADAM_B1);
      vv += (gr * gr - vv) * (1.f - ADAM_B2);
      mo[n] = mm; vo[n] = vv;
      th[n] = adam_step(x, mm * scal[alpha_idx], vv);
    } else {
      out[n] = t;
    }
  }
  if (th && sq_partials) {
    sq = wave_sum(sq);
    if (c == 0) sq_partials[blockIdx.x] = sq;
  }
}

// ---- recommend(): seen items -> -inf, then top-k by score (Base/BaseRecommender.py:189-234) ---------
// One workgroup per score row.  The row is staged in LDS when it fits (lds_cap floats), otherwise the
// selection works in place on the (private) score buffer.  k rounds of a block-wide arg-max with
// removal; ties go to the smaller item id; exhausted rows (everything -inf) yield -1.
__global__ __launch_bounds__(256) void mask_topk_kernel(float* __restrict__ scores, int ld, int W,
                                                        const int* __restrict__ row_ids,
                                                        const long long* __restrict__ seen_indptr,
                                                        const int* __restrict__ seen_indices, int k, int lds_cap,
                                                        int* __restrict__ out_items, float* __restrict__ out_vals,
                                                        const unsigned char* __restrict__ item_mask,
                                                        const long long* __restrict__ cold_indptr) {
  extern __shared__ __attribute__((aligned(16))) float srow[];
  __shared__ float wv[4];
  __shared__ int wi[4];
  const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float* g = scores + (size_t)r * ld;
  const bool in_lds = W <= lds_cap;
  float* row = in_lds ? srow : g;
  if (in_lds)
    for (int i = tid; i < W; i += 256) srow[i] = g[i];
  __syncthreads();
  // MF contract (Base/BaseMatrixFactorizationRecommender.py:113-119,128-143): items outside items_to_compute score -inf; a row
  // without a training interaction (cold) scores -inf everywhere
  const bool cold = cold_indptr != nullptr && cold_indptr[row_ids[r] + 1] == cold_indptr[row_ids[r]];
  if (item_mask != nullptr || cold)
    for (int i = tid; i < W; i += 256)
      if (cold || !item_mask[i]) row[i] = -INFINITY;
  if (seen_indptr) {
    const int u = row_ids[r];
    const long long s = seen_indptr[u], e = seen_indptr[u + 1];
    for (long long j = s + tid; j < e; j += 256) row[seen_indices[j]] = -INFINITY;
  }
  __syncthreads();
  for (int t = 0; t < k; ++t) {
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int i = tid; i < W; i += 256) {
      const float v = row[i];
      if (v > bv) { bv = v; bi = i; }       // strided scan visits ids in increasing order: first max wins
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(bv, o);
      const int oi = __shfl_xor(bi, o);
      if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    if (lane == 0) { wv[wave] = bv; wi[wave] = bi; }
    __syncthreads();
    if (tid == 0) {
      float v = wv[0]; int i = wi[0];
      for (int w = 1; w < 4; ++w)
        if (wv[w] > v || (wv[w] == v && wi[w] < i)) { v = wv[w]; i = wi[w]; }
      const bool ok = v > -INFINITY && i != 0x7fffffff;
      out_items[(size_t)r * k + t] = ok ? i : -1;
      out_vals[(size_t)r * k + t] = ok ? v : -INFINITY;
      if (ok) row[i] = -INFINITY;
    }
    __syncthreads();
  }
}

// ganmf_scores under a score filter (ganmf_set_score_filter): one workgroup per score row, same rule as mask_topk_kernel
__global__ __launch_bounds__(256) void score_filter_kernel(float* __restrict__ scores, int ld, int W, const int* __restrict__ row_ids,
                                                           const unsigned char* __restrict__ item_mask,
                                                           const long long* __restrict__ cold_indptr) {
  float* row = scores + (size_t)blockIdx.x * ld;
  const bool cold = cold_indptr != nullptr && cold_indptr[row_ids[blockIdx.x] + 1] == cold_indptr[row_ids[blockIdx.x]];
  if (item_mask == nullptr && !cold) return;
  for (int i = threadIdx.x; i < W; i += 256)
    if (cold || !item_mask[i]) row[i] = -INFINITY;
}

__global__ void fill_kernel(float* __restrict__ p, float v, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x)
    p[i] = v;
}

// Hold-out ranking metrics on the device (Base/Evaluation/Evaluator.py:262-335 with the metric definitions of
// Base/Evaluation/metrics.py: roc_auc, precision, precision_recall_min_denominator, recall, map, rr, ndcg, arhr and the
// hit count).  One THREAD per evaluated user: it walks the user's top-K list (mask_topk_kernel's output, -1 padded) once
// per cut-off, finds each recommended item in the user's sorted test row by binary search and forms the user's metric
// values in float64 exactly as ganmf_amd/evaluation.py::EvaluatorHoldoutFast does; the per-user values are summed per
// block in a fixed tree order and the host adds the block partials in block order (bitwise reproducible).
constexpr int EVAL_METRICS = 9;      // ROC_AUC, PRECISION, PRECISION_RECALL_MIN_DEN, RECALL, MAP, MRR, NDCG, HIT_RATE, ARHR
constexpr int EVAL_MAX_CUTOFFS = 8;
struct EvalP {
  const int* items;            // [n, K] recommended ids, -1 padded
  int K, n;
  const int* ids;              // [n] evaluated rows of the test matrix
  const long long* t_indptr;   // test matrix, evaluation orientation, column indices sorted inside a row
  const int* t_indices;
  const double* t_gain;        // 2^rating - 1 per stored entry
  const double* disc;          // [K] 1 / ln(rank + 1), as the host evaluator forms it (float32 logarithm)
  const double* ideal_cum;     // [n, K] prefix sums of the user's ideal DCG terms
  int ncut;
  int cutoffs[EVAL_MAX_CUTOFFS];
  double* partials;            // [gridDim.x][ncut][EVAL_METRICS]
};

__global__ __launch_bounds__(256) void eval_topk_kernel(const EvalP p) {
  __shared__ double red[256];
  const int u = blockIdx.x * 256 + threadIdx.x;
  const bool live = u < p.n;
  long long t0 = 0, t1 = 0;
  if (live) { const int r = p.ids[u]; t0 = p.t_indptr[r]; t1 = p.t_indptr[r + 1]; }
  const double n_test = (double)(t1 - t0);
  for (int ci = 0; ci < p.ncut; ++ci) {
    const int c = p.cutoffs[ci];
    double m[EVAL_METRICS];
#pragma unroll
    for (int q = 0; q < EVAL_METRICS; ++q) m[q] = 0.0;
    if (live) {
      double hits = 0.0, nneg = 0.0, len = 0.0, pairs = 0.0, ap = 0.0, arhr = 0.0, dcg = 0.0, rr = 0.0;
      for (int i = 0; i < c && i < p.K; ++i) {
        const int it = p.items[(size_t)u * p.K + i];
        if (it < 0) continue;
        len += 1.0;
        long long lo = t0, hi = t1;          // first stored index >= it
        while (lo < hi) {
          const long long mid = (lo + hi) >> 1;
          if (p.t_indices[mid] < it) lo = mid + 1; else hi = mid;
        }
        const bool hit = lo < t1 && p.t_indices[lo] == it;
        const double inv_rank = 1.0 / (double)(i + 1);
        if (hit) {
          hits += 1.0;
          ap += hits * inv_rank;
          arhr += inv_rank;
          if (rr == 0.0) rr = inv_rank;
          dcg += p.t_gain[lo] * p.disc[i];
        } else {
          nneg += 1.0;
          pairs += hits;                     // every hit ranked before this miss is a correctly ordered pair
        }
      }
      const double den = fmin(n_test, len);
      m[0] = nneg == 0.0 ? 1.0 : (hits > 0.0 ? pairs / (hits * nneg) : 0.0);
      m[1] = len > 0.0 ? hits / len : 0.0;
      m[2] = len > 0.0 ? hits / fmax(den, 1.0) : 0.0;
      m[3] = hits / n_test;
      m[4] = len > 0.0 ? ap / fmax(den, 1.0) : 0.0;
      m[5] = rr;
      if (dcg > 0.0) {
        const double ideal = p.ideal_cum[(size_t)u * p.K + (len > 0.0 ? (int)len - 1 : 0)];
        m[6] = dcg / (ideal > 0.0 ? ideal : 1.0);
      }
      m[7] = hits;
      m[8] = arhr;
    }
#pragma unroll
    for (int q = 0; q < EVAL_METRICS; ++q) {
      red[threadIdx.x] = m[q];
      __syncthreads();
      for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
      }
      if (threadIdx.x == 0) p.partials[((size_t)blockIdx.x * p.ncut + ci) * EVAL_METRICS + q] = red[0];
      __syncthreads();
    }
  }
}

}  // namespace ganmf
