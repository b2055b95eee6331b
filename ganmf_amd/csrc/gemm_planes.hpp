// EXPERIMENT (round 4; compiled by `make DIAG=1` only; GANMF_TUNE planes=1 in the stand-alone GEMM entry; bit-identical to the in-loop
// split in 24 shape / layout cases) -- measured LEVEL with gemm_bf16k.hpp: -1 % .. -6 % per launch at the step's shapes (tools/planes_bench.py,
// profiles/r04_wgrad_stream.md): 48 KiB of planes per K-tile through the CU's global -> LDS path cost what the 44 VALU per thread of the
// in-loop split cost.  Not used by the training step.
//
// The 16-wave split-bf16 GEMM (gemm_bf16k.hpp) on operands that were split into their three bf16 pieces AHEAD of the launch
// (gfx950 / CDNA4): the K loop is LDS-DMA of piece planes, fragment reads and MFMAs -- no operand registers, no split VALU.
//
// When both operands of a product are frozen or freshly produced as planes: the generator pass of GANMF (GANMF.py:191-203 -- the
// discriminator's weights We_ext / Wd_ext do not change while the reference walks its generator updates, so their planes are made
// once per pass) with the activations filed as planes by the kernels that produce them (EpiD::planes: slab-sum kernel, GEMM
// epilogue, CSR row expansion).  Timing-only builds priced it (profiles/r04_wgrad_stream.md, addendum): the splits of BOTH operands
// out of the loop take 21-23 % off a launch, those of one operand 2-3 % (the A- and B-staging waves share every SIMD and barrier).
//
//   * same tile (64 x 64 per 1024-thread workgroup), same four K groups, same chunk -> group deal, same six piece products in the
//     same order into the same two accumulators, same epilogue (gemm_epilogue, KG = 4) as bf16k_mainloop: results are bit-identical
//     to the in-loop split (the pieces are the same numbers: split_bf16x3 once per element instead of once per tile);
//   * planes are plain row-major bf16 matrices of the fp32 operand's geometry (piece q at base + q * stride).  K-contiguous
//     operands ([rows][K]) land in the LDS image of SplitStage<64, 64, false> (16-byte chunk XOR-swizzled by the row, applied to the
//     SOURCE address: the DMA writes lanes linearly) and are read with ds_read_b128; a K-major B ([K][N], the weights of the NN
//     products) lands as [64 k][64 n] rows with the chunk index XOR 4 on odd k-pairs and is read with the transposing
//     ds_read_b64_tr_b16 (two reads per piece fragment, conflict-free: the layout of wgrad_stream.hpp);
//   * a three-slot ring of 48 KiB stages (3 pieces x (A + B) x 8 KiB), three 1-KiB DMA pieces per wave and stage, counted vmcnt
//     waits, ONE barrier per K-tile; the ring is the epilogue's staging area afterwards.
#pragma once
#include "gemm_bf16s.hpp"

namespace ganmf {

constexpr int GPL_PIECE = 64 * 64 * 2;            // bytes of one piece plane of one operand tile: [64][64] bf16
constexpr int GPL_STAGE = 6 * GPL_PIECE;          // A hi, mid, lo, B hi, mid, lo
constexpr int GPL_NS = 3;

__device__ __forceinline__ void gpl_glds16(const void* src, unsigned lds_byte_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(src), "s"(lds_byte_addr) : "memory");
}
template <int N> __device__ __forceinline__ void gpl_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <bool BKM>
__device__ __forceinline__ void gemm_planes_body(const GemmP& p, const int bid, const int nblk, float* __restrict__ smem) {
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kg = wave >> 2;
  const int wr = (wave >> 1) & 1, wc = wave & 1;
  const int li = lane & 31, lh = lane >> 5;
  int tm, tn, sp, bz;
  tile_coords(p, bid, nblk, tm, tn, sp, bz);
  const int m0 = tm * 64, n0 = tn * 64;
  const int kbeg = sp * p.k_per_split;
  const int kend = min(p.K, kbeg + p.k_per_split);
  const int nt = (kend - kbeg + 63) / 64;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) float*)smem;
  const unsigned char* const zp = reinterpret_cast<const unsigned char*>(p.zero_page) + lane * 16;

  // ---- this wave's three DMA pieces of a stage: J = wave + 16 i; J < 24: piece J / 8 of A, rows 8 (J % 8) .. + 7; else of B
  const unsigned char* src[3];
  long long step[3];
  int krow[3];      // K-major B only: this lane's k inside the K-tile (row validity is checked per tile); -1: never valid
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int J = wave + 16 * i;
    const bool isA = J < 24;
    const int q = (isA ? J : J - 24) >> 3, sub = J & 7;
    krow[i] = -2;      // K-contiguous: validity is fixed (pointer or zero page)
    if (isA || !BKM) {
      const int row = 8 * sub + (lane >> 3), pos = lane & 7;
      const int c8 = pos ^ ((row >> 1) & 7);                       // the chunk that belongs at swizzled position `pos` (SplitStage<64, 64, false>)
      const int grow = (isA ? m0 : n0) + row;
      const bool ok = grow < (isA ? p.M : p.N);
      const bf16raw* base = isA ? p.a_planes + (size_t)q * p.a_pstride + (size_t)bz * p.a_batch_stride
                                : p.b_planes + (size_t)q * p.b_pstride;
      const int ld = isA ? p.lda : p.ldb;
      src[i] = ok ? reinterpret_cast<const unsigned char*>(base + (size_t)grow * ld + kbeg + 8 * c8) : zp;
      step[i] = ok ? 128 : 0;
    } else {
      const int k = 8 * sub + (lane >> 3), pos = lane & 7;
      const int c = pos ^ (((k >> 1) & 1) << 2);
      const int col = n0 + 8 * c;
      krow[i] = col < p.ldb ? k : -1;
      src[i] = reinterpret_cast<const unsigned char*>(p.b_planes + (size_t)q * p.b_pstride + (size_t)(kbeg + k) * p.ldb + col);
      step[i] = (long long)64 * p.ldb * 2;
    }
  }
  int issued = 0;
  auto issue_stage = [&]() {
    const unsigned slot = lds0 + (unsigned)(issued % GPL_NS) * GPL_STAGE;
    const int kleft = kend - kbeg - 64 * issued;      // k's from this tile's first to the end of the range
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const bool ok = krow[i] == -2 || (krow[i] >= 0 && krow[i] < kleft);
      gpl_glds16(ok ? src[i] : zp, slot + (unsigned)(wave + 16 * i) * 1024u);
      src[i] += step[i];
    }
    ++issued;
  };
  if (nt > 0) issue_stage();
  if (nt > 1) issue_stage();

  // ---- fragment addresses inside a stage
  using FA = SplitStage<64, 64, false>;
  const int l16 = lane & 15, qq = l16 >> 2, pp = l16 & 3, gi = (lane >> 4) & 1;
  const unsigned btr_off = (unsigned)(3 * GPL_PIECE + (8 * lh + qq) * 128 + (((wc * 4 + gi * 2 + (pp >> 1)) ^ ((qq >> 1) << 2)) * 16) + (pp & 1) * 8);
  typedef __attribute__((address_space(3))) s16x4* lds_s16x4_p;

  f32x16 acc, accl;
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc[r] = 0.f; accl[r] = 0.f; }

  for (int t = 0; t < nt; ++t) {
    if (issued - t >= 2) gpl_wait_vm<3>(); else gpl_wait_vm<0>();      // this wave's pieces of stage t have landed ...
    __builtin_amdgcn_s_barrier();                                        // ... everybody's; and stage t - 1 has been read by all
    if (issued < nt) issue_stage();
    const unsigned sl = lds0 + (unsigned)(t % GPL_NS) * GPL_STAGE;
    const unsigned* const pl = reinterpret_cast<const unsigned*>(smem) + (size_t)(t % GPL_NS) * (GPL_STAGE / 4);
    bf16x8 pa[3], pb[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) pa[q] = __builtin_bit_cast(bf16x8, FA::frag(pl + q * (GPL_PIECE / 4), wr * 32, kg, li, lh));
    if constexpr (!BKM) {
#pragma unroll
      for (int q = 0; q < 3; ++q) pb[q] = __builtin_bit_cast(bf16x8, FA::frag(pl + (3 + q) * (GPL_PIECE / 4), wc * 32, kg, li, lh));
    } else {
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const unsigned a = sl + btr_off + q * GPL_PIECE + kg * 16 * 128;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)a);
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(a + 4 * 128));
        pb[q] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
      }
    }
    // (mid,mid) (hi,lo) (lo,hi) (mid,hi) (hi,mid) -> accl, (hi,hi) -> acc: the order of bf16k_mainloop / gemm_bf16s_body
    accl = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[1], pb[1], accl, 0, 0, 0);
    accl = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[0], pb[2], accl, 0, 0, 0);
    accl = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[2], pb[0], accl, 0, 0, 0);
    accl = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[1], pb[0], accl, 0, 0, 0);
    accl = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[0], pb[1], accl, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[0], pb[0], acc, 0, 0, 0);
  }
  acc += accl;
  gpl_wait_vm<0>();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();      // the ring is idle: it becomes the epilogue's staging area
  f32x16 out[1][1];
  out[0][0] = acc;
  static_assert(4 * 64 * 64 * 4 <= GPL_NS * GPL_STAGE, "the ring must hold the four staged partial tiles");
  gemm_epilogue<64, 64, 1, 1, 4>(p, out, smem, TileCoord{tm, tn, sp, bz, m0, n0});
}

template <bool BKM>
__global__ __launch_bounds__(1024) void gemm_planes_mfma(const GemmP p) {
  __shared__ __attribute__((aligned(1024))) float smem[GPL_NS * GPL_STAGE / 4];      // 144 KiB: one workgroup per CU
  gemm_planes_body<BKM>(p, (int)blockIdx.x, (int)gridDim.x, smem);
}

inline hipError_t gemm_planes_launch(hipStream_t st, const GemmP& p, bool bkm) {
  const int grid = p.tiles_m * p.tiles_n * p.nsplit * p.nbatch;
  if (grid <= 0) return hipSuccess;
  if (bkm) GANMF_LAUNCH((gemm_planes_mfma<true>), dim3(grid), dim3(1024), 0, st, p);
  else GANMF_LAUNCH((gemm_planes_mfma<false>), dim3(grid), dim3(1024), 0, st, p);
  return hipGetLastError();
}

// planes of a dense fp32 range: n4 float4 from src -> three bf16 ranges at dst, dst + pstride, dst + 2 pstride (element offsets equal)
__global__ __launch_bounds__(256) void planes_of_kernel(const float* __restrict__ src, bf16raw* __restrict__ dst, long long pstride, long long n4) {
  const PlaneRef pl{dst, pstride};
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const float4 v = *reinterpret_cast<const float4*>(src + 4 * i);
    planes_store4(pl, (size_t)(4 * i), v.x, v.y, v.z, v.w);
  }
}

}  // namespace ganmf
