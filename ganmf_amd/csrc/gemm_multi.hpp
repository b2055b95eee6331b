// Launches that carry more than one piece of the GANMF step.
//
// Why: on this eight-XCD part every kernel boundary of a dependent chain costs ~4.4 us before the next kernel's first
// wave runs (rocprofv3 shows d_coef, densify and the split-K reduce -- a few MB of traffic each -- at 4.4-4.8 us; the
// per-XCD L2s are written back and invalidated at the boundary), and a D+G pair of the C2 step has 24 of them
// (profiles/r02_step_classes.md).  Where two neighbours of the chain do not depend on each other they ride in ONE launch,
// as block ranges of the same grid running different bodies:
//   front_kernel       generator GEMM F = U[uids].V^T (rows gathered by the operand fetch, GemmP::a_gather) + the CSR row
//                      expansion of the real rows (kernels.hpp densify_row_body): D- and G-step           (GANMF.py:82-83,183-184)
//   pair_kernel        gUb = dF.V (split-K slabs, summed later by adam_rows_kernel) + gV = dF^T.Ub with the fused Adam
//                      epilogue: the update of V is written to a SECOND buffer (EpiD::adam_theta_out) because gUb reads
//                      the old V inside the same launch; the host swaps the two buffers                   (GANMF.py:139)
//   de_dcoef_kernel    dE = Delta.Wd^T as split-K slabs + the hinge scalars / row scales / Es = rs (.) E of the same step:
//                      the slabs are scaled when they are summed, one launch later                        (GANMF.py:131-132)
//   gemm_bf16s_red     gWd_ext = Es^T.Delta with the fused Adam epilogue + the slab sum / row scale of dE = rs (.) (Delta.Wd^T),
//                      which the previous launch left as split-K slabs and only the NEXT launch (gWe) reads (GANMF.py:138)
//   wgrad_pair_kernel  gWd_ext + gWe_ext, both with the fused Adam epilogue, behind a stand-alone slab sum of dE: 2 088 workgroups
//                      where the chip holds 1 536, so the K phase of the second round runs under the Adam streams of the
//                      first (58.5 us as two launches with everything resident and in phase, 48.3 + 4.5 us this way)  (GANMF.py:138)
// The bodies are the ordinary kernels' bodies (gemm_f32_body, gemm_bf16s_body, splitk_reduce_body): same arithmetic, same
// summation order, bit-identical results to the separate launches.
#pragma once
#include "gemm_bf16s.hpp"
#include "gemm_bf16k.hpp"
#ifdef GANMF_PERSIST_DIAG_BUILD
#include "gemm_planes.hpp"      // experiment: the 16-wave plan on pre-split operand planes
#endif
#include "gemm_f32.hpp"
#include "kernels.hpp"
#include "gemm_skinny.hpp"
#include "gemm_bf16w.hpp"

namespace ganmf {

// blocks [0, ng): 64 x 64 tiles of the NT GEMM; blocks [ng, ng + d.nb): one CSR row each
// (X3: the 16-wave split-bf16 loop instead of the fp32 MFMA, gemm_bf16k.hpp -- whichever the stand-alone product would run)
template <int KG, bool X3 = false>
__global__ __launch_bounds__(256 * KG) void front_kernel(const GemmP g, const DensP d) {
  __shared__ __attribute__((aligned(16))) float smem[3 * (64 + 64) * 64];
  static_assert(!X3 || KG == 4, "the split-bf16 form has four K groups");
  const int ng = g.tiles_m * g.tiles_n * g.nsplit * g.nbatch;
  if ((int)blockIdx.x < ng) {
    if constexpr (X3) gemm_bf16k_body<false, false>(g, (int)blockIdx.x, ng, smem);
    else gemm_f32_body<64, 64, 64, 3, false, false, KG>(g, (int)blockIdx.x, ng, smem);
  } else densify_row_body(d, (int)blockIdx.x - ng);
}
// ... and on a handle whose products run in ONE low-precision piece (mfma = "f16" | "bf16": BASELINE configs[4] as written): the 16-wave
// one-piece loop the stand-alone generator product runs (gemm_bf16k_mfma<false, false, 1, F16>), 64 KiB of LDS
template <bool F16>
__global__ __launch_bounds__(1024) void front_lp_kernel(const GemmP g, const DensP d) {
  __shared__ __attribute__((aligned(16))) float smem[bf16k_smem_dw<1>()];
  const int ng = g.tiles_m * g.tiles_n * g.nsplit * g.nbatch;
  if ((int)blockIdx.x < ng) gemm_bf16k_body<false, false, 1, F16>(g, (int)blockIdx.x, ng, smem);
  else densify_row_body(d, (int)blockIdx.x - ng);
}

// blocks [0, ng): 64 x 64 tiles of the NT GEMM dE = Delta . Wd^T, which here only writes its split-K slabs; blocks [ng, ng + nd):
// the discriminator scalars and Es = rs (.) E (kernels.hpp d_coef_body), whose outputs the GEMM does not read -- the row scale
// of dE is applied by the slab sum, in the NEXT launch
template <int KG, bool X3 = false>
__global__ __launch_bounds__(256 * KG) void de_dcoef_kernel(const GemmP g, const DCoefP d, const int nd) {
  __shared__ __attribute__((aligned(16))) float smem[3 * (64 + 64) * 64];
  static_assert(!X3 || KG == 4, "the split-bf16 form has four K groups");
  const int ng = g.tiles_m * g.tiles_n * g.nsplit * g.nbatch;
  if (nd == 0) {
    // every GEMM workgroup first does its 1 / ng share of d_coef (a few hundred float4 of Es; the two scalar sums are formed
    // by each workgroup alike, block 0 publishes them): extra d_coef workgroups would need a 96-KiB slot of their own and run
    // as a tail behind the GEMM's (24.8 vs 21.6 us for the same product without them)
    d_coef_body(d, (int)blockIdx.x, ng, smem);
    __syncthreads();
    if constexpr (X3) gemm_bf16k_body<false, false>(g, (int)blockIdx.x, ng, smem);
    else gemm_f32_body<64, 64, 64, 3, false, false, KG>(g, (int)blockIdx.x, ng, smem);
    return;
  }
  if ((int)blockIdx.x < ng) {
    if constexpr (X3) gemm_bf16k_body<false, false>(g, (int)blockIdx.x, ng, smem);
    else gemm_f32_body<64, 64, 64, 3, false, false, KG>(g, (int)blockIdx.x, ng, smem);
  } else d_coef_body(d, (int)blockIdx.x - ng, nd, smem);
}

// blocks [0, n0): g0, an NN GEMM; blocks [n0, n0 + n1): g1, a TN GEMM
// NS = 2: 64 KiB of LDS and 52 VGPRs per workgroup, so TWO of the 16-wave workgroups share a CU and the 464 workgroups of
// the C2 pair are resident in one round (with the 96 KiB ring of the single-GEMM launches they ran as two rounds)
// SGPRs capped at 80 (round 6): a CU admits floor(800 / (ceil(sgpr / 16) * 16 + 16)) waves per SIMD -- at the 106 scalar registers hipcc took for this
// kernel that is 6, i.e. ONE 16-wave workgroup per CU whatever its LDS, and in-kernel stamps showed the gV range entering only after the gUb range had
// left (profiles/r06_pair_kernel.md); at <= 80 it is 8, two workgroups per CU, which is what NS = 2 was built for.
template <int KG, int NS>
__global__ __launch_bounds__(256 * KG) __attribute__((amdgpu_num_sgpr(80))) void pair_kernel(const GemmP g0, const GemmP g1) {
  __shared__ __attribute__((aligned(16))) float smem[NS * (64 + 64) * 64];
  const int n0 = g0.tiles_m * g0.tiles_n * g0.nsplit * g0.nbatch;
  const int n1 = g1.tiles_m * g1.tiles_n * g1.nsplit * g1.nbatch;
  if ((int)blockIdx.x < n0) gemm_f32_body<64, 64, 64, NS, false, true, KG>(g0, (int)blockIdx.x, n0, smem);
  else gemm_f32_body<64, 64, 64, NS, true, true, KG>(g1, (int)blockIdx.x - n0, n1, smem);
}

// the same pair on a low-precision handle: gUb (NN, split-K slabs) and gV (TN, fused Adam into the second V buffer) on the 16-wave one-piece
// loop each of them runs stand-alone (gemm_bf16k_mfma<.., 1, F16>: bit-identical); 64 KiB of LDS, two workgroups per CU
template <bool F16>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_num_sgpr(80))) void pair_lp_kernel(const GemmP g0, const GemmP g1) {
  __shared__ __attribute__((aligned(16))) float smem[bf16k_smem_dw<1>()];
  const int n0 = g0.tiles_m * g0.tiles_n * g0.nsplit * g0.nbatch;
  const int n1 = g1.tiles_m * g1.tiles_n * g1.nsplit * g1.nbatch;
  if ((int)blockIdx.x < n0) gemm_bf16k_body<false, true, 1, F16>(g0, (int)blockIdx.x, n0, smem);
  else gemm_bf16k_body<true, true, 1, F16>(g1, (int)blockIdx.x - n0, n1, smem);
}

// blocks [0, ng): 64 x 64 x 32 tiles of the TN split-bf16 GEMM; blocks [ng, ng + nred): slab reduce of ANOTHER product
__global__ __launch_bounds__(256, 2) void gemm_bf16s_red(const GemmP g, const RedP r, const int nred) {
  __shared__ __attribute__((aligned(16))) float smem[Bf16sLds<64, 64, 32, 3>::DW];
  const int ng = g.tiles_m * g.tiles_n * g.nsplit * g.nbatch;
  if ((int)blockIdx.x < ng) gemm_bf16s_body<64, 64, 32, true, true, 3>(g, (int)blockIdx.x, ng, smem);
  else splitk_reduce_body(r, (int)blockIdx.x - ng, nred, 0, smem);
}

// blocks [0, ng): 64 x 64 tiles of the TN staged GEMM with the fused Adam epilogue (DisGANMF: gW_0_ext = [X;F | 1 (| uid)]^T . dz_0 + Adam(W_0));
// blocks [ng, ng + ncs): the column sums of the top of the backward pass -- output-layer gradient + Adam(w_o), and in the low-precision modes the
// fp32 float(uid) row of W_0_ext (kernels.hpp dis_colsum_body) -- which read only what the launch in front wrote (layer output, dlogit, dz_0)
// and write tensors the GEMM does not touch: dis_dz_top_kernel's launch (7 us of a 46 us discriminator step at configs[4]) leaves the step.
template <int BK, int NPIECE, bool F16>
__global__ __launch_bounds__(256, 2) void gemm_bf16s_colsum(const GemmP g, const ColSumP q, const int ncs) {
  __shared__ __attribute__((aligned(16))) float smem[Bf16sLds<64, 64, BK, NPIECE>::DW];
  static_assert(Bf16sLds<64, 64, BK, NPIECE>::DW >= 2 * CS_GROUPS * CS_COLS, "the column sums' two LDS images");
  const int ng = g.tiles_m * g.tiles_n * g.nsplit * g.nbatch;
  if ((int)blockIdx.x < ng) gemm_bf16s_body<64, 64, BK, true, true, NPIECE, F16>(g, (int)blockIdx.x, ng, smem);
  else dis_colsum_body(q, (int)blockIdx.x - ng, smem);
}

// blocks [0, n0): g0; blocks [n0, n0 + n1): g1 -- two independent 64 x 64 x 32 TN split-bf16 products with the fused Adam
// epilogue (gWd_ext and gWe_ext of the discriminator step).  The launch has more workgroups than the chip holds at once, so
// the K phase of the later workgroups runs under the Adam streams of the earlier ones.
__global__ __launch_bounds__(256, 2) void wgrad_pair_kernel(const GemmP g0, const GemmP g1) {
  __shared__ __attribute__((aligned(16))) float smem[Bf16sLds<64, 64, 32, 3>::DW];
  const int n0 = g0.tiles_m * g0.tiles_n * g0.nsplit * g0.nbatch;
  const int n1 = g1.tiles_m * g1.tiles_n * g1.nsplit * g1.nbatch;
  if ((int)blockIdx.x < n0) gemm_bf16s_body<64, 64, 32, true, true, 3>(g0, (int)blockIdx.x, n0, smem);
  else gemm_bf16s_body<64, 64, 32, true, true, 3>(g1, (int)blockIdx.x - n0, n1, smem);
}

// (round 6, GANMF_TUNE wgrad_seam) the same two products with the slab sum of dE -- which only the gWe range reads -- as the FIRST block range of the launch instead of a
// launch of its own: blocks [0, nred) sum the slabs and store dE write-through (sc1), drain and count themselves in on an agent-scope counter; blocks [nred, nred + n0) are
// gWd_ext + Adam(Wd), which never read dE; blocks [nred + n0, ...) are gWe_ext + Adam(We): one lane polls the counter (relaxed), one agent-scope acquire, barrier, then the
// ordinary body (cdna guide section 6, Guideline 16, counter form).  Workgroups are dispatched in block order, so every reduce block has been DISPATCHED before the first gWe
// block exists, and by the time the chip has worked through 512 reduce and 928 gWd blocks the sums have long landed: the poll normally passes at once.  Liveness does not rest on
// that alone: the poll is bounded, and a block that gives up raises cnt[1] (the host turns it into an error) and goes on rather than hang.
__global__ __launch_bounds__(256, 2) void wgrad_seam_kernel(const GemmP g0, const GemmP g1, const RedP r, const int nred, unsigned long long* cnt,
                                                            const unsigned long long target) {
  __shared__ __attribute__((aligned(16))) float smem[Bf16sLds<64, 64, 32, 3>::DW];
  static_assert(Bf16sLds<64, 64, 32, 3>::DW >= 4 + 4 * 256, "splitk_reduce_body's LDS");
  const int n0 = g0.tiles_m * g0.tiles_n * g0.nsplit * g0.nbatch;
  const int n1 = g1.tiles_m * g1.tiles_n * g1.nsplit * g1.nbatch;
  const int b = (int)blockIdx.x;
  if (b < nred) {
    splitk_reduce_body(r, b, nred, 0, smem);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // EVERY storing wave drains its write-through stores ...
    __syncthreads();                                       // ... before ONE lane counts the block in
    if (threadIdx.x == 0) __hip_atomic_fetch_add(cnt, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  if (b < nred + n0) { gemm_bf16s_body<64, 64, 32, true, true, 3>(g0, b - nred, n0, smem); return; }
  if (threadIdx.x == 0) {
    unsigned spins = 0;
    while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(4);
      if (++spins > (1u << 22)) { __hip_atomic_store(cnt + 1, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }      // (seconds: something is badly wrong)
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  gemm_bf16s_body<64, 64, 32, true, true, 3>(g1, b - nred - n0, n1, smem);
}

// ---- host side: can this plan ride in the combined launch?
inline bool plan_is_f32_64_kg(const GemmPlan& pl, int kg) {
  return pl.mode == MFMA_F32 && pl.tile == 64 && pl.ring == 3 && pl.kg == kg && !pl.persist;
}
// ... or would the stand-alone product run the 16-wave ONE-piece loop (a forced bf16 / fp16 handle, GANMF_X3KG bit 2: plan_gemm)?
inline bool plan_is_lp_64(const GemmPlan& pl) {
  return (pl.mode == MFMA_F16 || pl.mode == MFMA_BF16) && pl.tile == 64 && pl.ring == 3 && pl.bk != 32 && !pl.persist;
}

inline void fill_plan(GemmP& p, const GemmPlan& pl) {
  if (p.nbatch < 1) p.nbatch = 1;
  p.tiles_m = pl.tiles_m; p.tiles_n = pl.tiles_n;
  p.nsplit = pl.nsplit; p.k_per_split = pl.kps;
  p.epi.sq_stride = pl.sq_count;
  p.counters = nullptr;
}

}  // namespace ganmf
