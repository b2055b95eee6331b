// EXPERIMENT (round 4; compiled by `make DIAG=1` only, GANMF_WGRAD_STREAM=1 selects it; bit-identical to wgrad_pair_kernel, the
// parity suite passes with it as the default) -- measured SLOWER than the per-tile launch: 59 us against 48.8 us at configs[1].
// What the in-kernel stamps say and why it stays out of the product: profiles/r04_wgrad_stream.md.
//
// The discriminator step's two weight-gradient products with TF-Adam applied in place (GANMF.py:138: opt_disc.minimize),
// as ONE persistent launch whose HBM streams and matrix work overlap INSIDE a workgroup (gfx950 / CDNA4).
//
//   gWd_ext = Es^T . Delta   [e+1, N]      gWe_ext = [X;F|1]^T . dE   [N+1, e]        K = 2B batch rows for both
//
// Why a new structure (profiles/r03_gemm_stamps.md): the per-tile form (wgrad_pair_kernel, gemm_multi.hpp) is two ~30 us phases
// per workgroup -- a K loop bound by its own instruction stream (11 VALU per element pair for the exact 3-way split, 48
// ds_read_b32 per K-major fragment image) and the twelve Adam streams at HBM rate -- that overlap only across workgroups: 48 us
// for 186 MB.  Here
//   * the four activation operands are split into their bf16 x 3 planes ONCE per step (presplit_red_kernel below: one launch
//     together with the slab sum of dE that precedes this kernel anyway), as plain row-major [2B][ld] bf16 matrices;
//   * one 1024-thread workgroup per CU walks 64 x 128 gradient tiles.  Waves 0-7 (GEMM waves, one 32 x 32 block each) pull
//     32-row K stages of the planes straight into LDS with global_load_lds_dwordx4 (no registers, no VALU; a three-slot ring
//     that runs ACROSS tiles), read the K-major image with the transposing ds_read_b64_tr_b16 (a quarter of the LDS
//     instructions of the ds_read_b32 form, conflict-free through an XOR of the 16-byte chunk index on the source side) and
//     issue the same six piece products per 16-wide chunk into the same two accumulators as gemm_bf16s_body: every gradient
//     element is the SAME fp32 sum in the SAME order, bit-identical to the per-tile kernel;
//   * waves 8-15 (Adam waves) own the tile finished one step earlier: its gradient sits in a 32 KiB LDS tile, theta / m / v
//     arrive through a four-slot register ring that was requested a whole tile earlier (72-96 KiB in flight per CU), TF
//     ApplyAdam, three 16-byte stores per lane.  They share the GEMM waves' stage barriers, so the two halves of the workgroup
//     move in lockstep, one tile apart, and the HBM streams never stop while the matrix cores work.
// The sum(theta^2) partials of the L2 term are filed one per 64 x 128 tile (the per-epoch finish kernel sums whatever a step
// left in its arena segment).
#pragma once
#include <algorithm>
#include <vector>
#include "gemm_f32.hpp"

namespace ganmf {

struct PlaneJob { const float* src; PlaneRef dst; long long n4; };      // n4 float4 of a dense [rows][ld] range
constexpr int PLANE_JOBS = 4;
struct PlaneJobs { PlaneJob j[PLANE_JOBS]; int count; };

__device__ inline void presplit_body(const PlaneJobs& js, const int bx, const int nbx) {
  for (int i = 0; i < js.count; ++i) {
    const PlaneJob& jb = js.j[i];
    for (long long idx = (long long)bx * 256 + threadIdx.x; idx < jb.n4; idx += (long long)nbx * 256) {
      const float4 v = *reinterpret_cast<const float4*>(jb.src + 4 * idx);
      planes_store4(jb.dst, (size_t)(4 * idx), v.x, v.y, v.z, v.w);
    }
  }
}

// blocks [0, nred): slab sum of dE with its row scale (the reduce that precedes the weight-gradient launch; it also files dE's
// planes, RedP::planes); blocks [nred, grid): the planes of the operands that are final by now (XF, Delta, Es)
struct RedPlanes { PlaneRef pl; int on; };
__global__ __launch_bounds__(256) void presplit_red_kernel(const RedP r, const RedPlanes rp, const int nred, const PlaneJobs js);

// ---- the persistent kernel -------------------------------------------------------------------------------------------------
constexpr int WGS_BM = 64, WGS_BN = 128, WGS_BK = 32, WGS_NS = 3;
constexpr int WGS_A_PIECE = WGS_BK * WGS_BM * 2;                      // bytes: [32 k][64] bf16
constexpr int WGS_B_PIECE = WGS_BK * WGS_BN * 2;                      //        [32 k][128] bf16
constexpr int WGS_STAGE = 3 * (WGS_A_PIECE + WGS_B_PIECE);            // 36 KiB
constexpr int WGS_PIECES = WGS_STAGE / 1024;                          // 36 wave-wide DMA instructions per stage
constexpr int WGS_CBUF = WGS_BM * WGS_BN * 4;                         // 32 KiB gradient tile
constexpr int WGS_LDS = WGS_NS * WGS_STAGE + WGS_CBUF + 64;
constexpr int WGS_GEMM_WAVES = 8, WGS_ADAM_WAVES = 8;

struct WgsProd {
  const bf16raw* a_pl;      // planes of A [K][lda] (hi, mid, lo at a_pl + q * a_ps): gradient row m = column m of A
  const bf16raw* b_pl;      // planes of B [K][ldb]: gradient column n = column n of B
  long long a_ps, b_ps;
  int lda, ldb, ldc;
  int M, N, K;
  int tiles_m, tiles_n;     // 64 x 128 tiles, tile id = tm * tiles_n + tn (column fastest: neighbours share DRAM pages of a row)
  EpiD epi;                 // EPI_ADAM: adam_theta / adam_theta_out / adam_m / adam_v / adam_alpha / adam_reg, sq_partials, sp_rows
};
struct WgsP {
  WgsProd g[2];
  int tiles0, tiles_total;
  const int* table;         // tile schedule [rounds][gridDim.x]
  int rounds;
  const float* zero_page;
  float* dump;              // >= 8 KiB nobody reads: the Adam waves' lanes outside the matrix store there (no branch around a store:
                            // every iteration issues the same vector-memory operations, so hipcc's counted waits are exact)
  int diag;                 // diagnostic builds only (make DIAG=1; wrong results): 1 no Adam streams, 2 no operand DMA, 4 no fragment reads / MFMAs
  unsigned long long* stamps;   // diagnostic builds only: [workgroup][role 0 GEMM / 1 Adam][32] s_memrealtime stamps (100 MHz)
};
#ifdef GANMF_PERSIST_DIAG_BUILD
#define WGS_STAMP(role, i) do { if (p.stamps && lane == 0 && (wave == 0 || wave == WGS_GEMM_WAVES) && (i) < 32) \
    p.stamps[((size_t)blockIdx.x * 2 + (role)) * 32 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define WGS_STAMP(role, i) do { } while (0)
#endif

// Tile schedule (host-built, wgs_build_schedule below): table[i * nwg + wg] = the i-th tile of workgroup wg (-1: none; a workgroup's
// tiles are its first entries).  The 32 workgroups of an XCD (wg & 7: observed round-robin placement; speed only) work on a
// 4 x 8 block of tiles at a time: four A panels and eight B panels (1.9 MB of planes) serve 32 tiles out of the XCD's 4 MiB L2.
// In list order -- 32 different B panels per XCD and round -- the planes were re-fetched from the Infinity Cache for every tile:
// 267 MB per launch beside 176 MB of Adam streams, and the launch was bound by that traffic (profiles/r04_wgrad_stream.md).
__device__ inline int wgs_tile_of(const int* __restrict__ table, int wg, int nwg, int i) { return table[i * nwg + wg]; }

__device__ __forceinline__ void wgs_glds16(const void* src, unsigned lds_byte_addr) {
  // global_load_lds_dwordx4: lane L's 16 bytes land at M0 + 16 L (asm: invisible to hipcc's wait insertion, which would otherwise
  // treat the pending DMA as a flat access and drain lgkmcnt before every fragment read; gemm_f32.hpp issue_piece_asm)
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(src), "s"(lds_byte_addr) : "memory");
}

template <int N> __device__ __forceinline__ void wgs_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

__global__ __launch_bounds__(1024) void wgrad_stream_kernel(const WgsP p) {
  __shared__ __attribute__((aligned(1024))) unsigned char lds[WGS_LDS];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wg = (int)blockIdx.x, nwg = (int)gridDim.x;
  int ntile = 0;      // this workgroup's tiles: entries 0 .. ntile-1 of its column of the schedule
  while (ntile < p.rounds && wgs_tile_of(p.table, wg, nwg, ntile) >= 0) ++ntile;
  float* const cbuf = reinterpret_cast<float*>(lds + WGS_NS * WGS_STAGE);
  float* const sqbuf = cbuf + WGS_BM * WGS_BN;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;

  auto decode = [&](int id, int& pi, int& tm, int& tn) {
    pi = id >= p.tiles0 ? 1 : 0;
    const int lid = id - pi * p.tiles0;
    const int tnn = p.g[pi].tiles_n;
    tm = lid / tnn; tn = lid - tm * tnn;
  };

  // Roles by wave < 8: every SIMD hosts two GEMM and two Adam waves.  (Roles by wave & 2 -- GEMM waves on two SIMDs, Adam waves on
  // the other two -- measured 70 us against 59: half the matrix pipes idle.)
  const bool gemm_role = wave < WGS_GEMM_WAVES;
  const int role_idx = wave & (WGS_GEMM_WAVES - 1);      // 0 .. 7 inside the role
  if (gemm_role) {
    // ================================================= GEMM waves =================================================
    const int gw = role_idx;
    const int wr = gw >> 2, wc = gw & 3;
    const int li = lane & 31, lh = lane >> 5;
    const int np = gw < WGS_PIECES - 4 * WGS_GEMM_WAVES ? 5 : 4;      // DMA pieces of a stage issued by this wave: J = gw + 8 i < 36
    const unsigned char* const zp = reinterpret_cast<const unsigned char*>(p.zero_page) + lane * 16;

    // ---- DMA cursor: (round, stage) of the next stage to request; runs two stages ahead of the multiply
    int d_round = 0, d_stage = 0, d_nst = 0, d_K = 0;
    bool d_live = ntile > 0;
    const unsigned char* d_src[5];      // this lane's source of piece i at the cursor's stage
    long long d_step[5];                // bytes per stage
    int d_k[5];                         // this lane's row inside a stage (-1: column out of range, zero page)
    auto cursor_tile = [&]() {
      int pi, tm, tn;
      decode(wgs_tile_of(p.table, wg, nwg, d_round), pi, tm, tn);
#ifdef GANMF_PERSIST_DIAG_BUILD
      if (p.diag & 8) { tm = 0; tn = 0; }      // timing only: every workgroup fetches the operands of tile (0, 0) (all L2 hits)
#endif
      const WgsProd& g = p.g[pi];
      d_K = g.K;
      d_nst = (g.K + WGS_BK - 1) / WGS_BK;
#pragma unroll
      for (int i = 0; i < 5; ++i) {
        const int J = gw + 8 * i;
        if (J < 12) {             // A piece q, rows 8 sub .. 8 sub + 7 of the stage: lane -> (row, 16-byte chunk) of the linear LDS image
          const int q = J >> 2, sub = J & 3;
          const int k = 8 * sub + (lane >> 3), cs = lane & 7;
          const int c = cs ^ (((k >> 1) & 1) << 2);      // the chunk that belongs at swizzled position cs
          const int col = tm * WGS_BM + 8 * c;
          d_k[i] = col < g.lda ? k : -1;
          d_src[i] = reinterpret_cast<const unsigned char*>(g.a_pl + (size_t)q * g.a_ps + (size_t)k * g.lda + col);
          d_step[i] = (long long)WGS_BK * g.lda * 2;
        } else {                  // B piece q, rows 4 sub .. 4 sub + 3
          const int Jb = J - 12;
          const int q = Jb >> 3, sub = Jb & 7;
          const int k = 4 * sub + (lane >> 4), cs = lane & 15;
          const int c = cs ^ ((k & 3) << 2);
          const int col = tn * WGS_BN + 8 * c;
          d_k[i] = col < g.ldb ? k : -1;
          d_src[i] = reinterpret_cast<const unsigned char*>(g.b_pl + (size_t)q * g.b_ps + (size_t)k * g.ldb + col);
          d_step[i] = (long long)WGS_BK * g.ldb * 2;
        }
      }
    };
    int issued = 0;      // stages requested so far (ring slot = index % 3)
    auto issue_stage = [&]() {
      const unsigned slot = lds0 + (unsigned)(issued % WGS_NS) * WGS_STAGE;
      const int k0 = d_stage * WGS_BK;
#pragma unroll
      for (int i = 0; i < 5; ++i) {
#ifdef GANMF_PERSIST_DIAG_BUILD
        if (p.diag & 2) continue;
#endif
        if (i < 4 || np == 5) {
          const bool ok = d_k[i] >= 0 && k0 + d_k[i] < d_K;
          wgs_glds16(ok ? d_src[i] : zp, slot + (unsigned)(gw + 8 * i) * 1024u);
          d_src[i] += d_step[i];
        }
      }
      ++issued;
      if (++d_stage == d_nst) {
        d_stage = 0;
        if (++d_round < ntile) cursor_tile(); else d_live = false;
      }
    };
    if (d_live) cursor_tile();
    if (d_live) issue_stage();
    if (d_live) issue_stage();

    // ---- fragment addresses (bytes inside a stage).  ds_read_b64_tr_b16: lane 4 qq + pp of a 16-lane group supplies row qq,
    // columns 4 pp .. 4 pp + 3 of a 4 x 16 block and receives column (lane & 15) of the four rows.  Group (gi, h): operand rows
    // 16 gi .. 16 gi + 15 of the wave's 32, k = 8 h + 0 .. 7 of the 16-wide chunk in two reads (k + 0..3, k + 4..7).
    const int l16 = lane & 15, qq = l16 >> 2, pp = l16 & 3, gi = (lane >> 4) & 1;
    const unsigned a_off = (unsigned)((8 * lh + qq) * (WGS_BM * 2) + (((wr * 4 + gi * 2 + (pp >> 1)) ^ ((qq >> 1) << 2)) * 16) + (pp & 1) * 8);
    const unsigned b_off = (unsigned)(3 * WGS_A_PIECE + (8 * lh + qq) * (WGS_BN * 2) + (((wc * 4 + gi * 2 + (pp >> 1)) ^ (qq << 2)) * 16) + (pp & 1) * 8);
    typedef __attribute__((address_space(3))) s16x4* lds_s16x4_p;
    auto frag = [&](unsigned byte_addr, unsigned half_step) -> bf16x8 {
      const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)byte_addr);
      const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(byte_addr + half_step));
      const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
      return __builtin_bit_cast(bf16x8, v);
    };

    int done = 0;      // stages multiplied so far
    WGS_STAMP(0, 0);
    for (int r = 0; r < ntile; ++r) {
      int pi, tm, tn;
      decode(wgs_tile_of(p.table, wg, nwg, r), pi, tm, tn);
      const int nst = (p.g[pi].K + WGS_BK - 1) / WGS_BK;
      f32x16 acc, accl;
#pragma unroll
      for (int i = 0; i < 16; ++i) { acc[i] = 0.f; accl[i] = 0.f; }
      for (int s = 0; s < nst; ++s) {
        // this wave's pieces of stage `done` have landed once at most the pieces of the one younger stage are outstanding
        if (issued - done >= 2) { if (np == 5) wgs_wait_vm<5>(); else wgs_wait_vm<4>(); }
        else wgs_wait_vm<0>();
        __builtin_amdgcn_s_barrier();      // ... and every other wave's; and everybody has read the slot the next request overwrites
        if (r == 0) WGS_STAMP(0, 20 + s);
        const unsigned sl = lds0 + (unsigned)(done % WGS_NS) * WGS_STAGE;
        // all 24 fragment reads of the stage go out first, the next stage's DMA requests (address arithmetic and five issues) run
        // under their latency, then the twelve MFMAs
        bf16x8 pa[2][3], pb[2][3];
#ifdef GANMF_PERSIST_DIAG_BUILD
        if (!(p.diag & 4))
#endif
#pragma unroll
        for (int c = 0; c < WGS_BK / 16; ++c) {
#pragma unroll
          for (int q = 0; q < 3; ++q) pa[c][q] = frag(sl + q * WGS_A_PIECE + a_off + c * 16 * (WGS_BM * 2), 4 * (WGS_BM * 2));
#pragma unroll
          for (int q = 0; q < 3; ++q) pb[c][q] = frag(sl + q * WGS_B_PIECE + b_off + c * 16 * (WGS_BN * 2), 4 * (WGS_BN * 2));
        }
        __builtin_amdgcn_sched_barrier(0);
        if (d_live) issue_stage();
        __builtin_amdgcn_sched_barrier(0);
#ifdef GANMF_PERSIST_DIAG_BUILD
        if (!(p.diag & 4))
#endif
#pragma unroll
        for (int c = 0; c < WGS_BK / 16; ++c) {
          // (mid,mid) (hi,lo) (lo,hi) (mid,hi) (hi,mid) -> accl, (hi,hi) -> acc: the order of gemm_bf16s_body
          accl = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[c][1], pb[c][1], accl, 0, 0, 0);
          accl = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[c][0], pb[c][2], accl, 0, 0, 0);
          accl = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[c][2], pb[c][0], accl, 0, 0, 0);
          accl = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[c][1], pb[c][0], accl, 0, 0, 0);
          accl = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[c][0], pb[c][1], accl, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[c][0], pb[c][0], acc, 0, 0, 0);
        }
        ++done;
      }
      acc += accl;
      WGS_STAMP(0, 1 + 3 * r);
      __builtin_amdgcn_s_barrier();        // Y: the Adam waves have finished with the previous gradient tile
      WGS_STAMP(0, 2 + 3 * r);
#pragma unroll
      for (int i = 0; i < 16; ++i)
        cbuf[(wr * 32 + (i & 3) + 8 * (i >> 2) + 4 * lh) * WGS_BN + wc * 32 + li] = acc[i];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();        // X: this tile's gradient is in LDS
      WGS_STAMP(0, 3 + 3 * r);
    }
    if (ntile > 0) {                       // the Adam waves' last tile: same barriers, nothing to do
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_s_barrier();
    }
    return;
  }

  // =================================================== Adam waves ===================================================
  const int aw = role_idx;
  const int at = aw * 64 + lane;      // 0 .. 511
  // quantum q (0, 1) of a tile = its rows 32 q .. 32 q + 31: float4 index q * 1024 + j * 512 + at (j = 0, 1) -> row 32 q + 16 j + (at >> 5),
  // columns 4 (at & 31) ..: a wave covers two 512-byte row segments per instruction.  Two float4 per stream and thread give the
  // compiler two independent IEEE sqrt / divide chains to interleave (the update is latency-bound, not issue-bound: stamps in
  // profiles/r04_wgrad_stream.md).
  const int c4 = at & 31, r16 = at >> 5;
  const float* const zpf = p.zero_page + (at & 255) * 4;
  float* const dumpf = p.dump + (size_t)at * 4;      // where the lanes outside the matrix put their (meaningless) results

  float4 T[2][2], Mo[2][2], V[2][2];      // the ring: slot q holds quantum q of the tile to update next
  auto request = [&](auto qc, int id) {      // theta / m / v of quantum q of tile `id`
    constexpr int q = decltype(qc)::value;
    int pi, tm, tn;
    decode(id, pi, tm, tn);
    const WgsProd& g = p.g[pi];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int row = tm * WGS_BM + q * 32 + j * 16 + r16, col = tn * WGS_BN + 4 * c4;
      const bool ok = row < g.M && col < g.N;
      const size_t off = (size_t)row * g.ldc + col;
      T[q][j] = *reinterpret_cast<const float4*>(ok ? g.epi.adam_theta + off : zpf);
      Mo[q][j] = *reinterpret_cast<const float4*>(ok ? g.epi.adam_m + off : zpf);
      V[q][j] = *reinterpret_cast<const float4*>(ok ? g.epi.adam_v + off : zpf);
    }
  };
  float sq = 0.f;
  float alpha = 0.f;
  // lr_t of the step, read ONCE: a load inside the iteration would be the youngest vector-memory operation when its value is
  // needed, and waiting for it would drain the whole request ring
  const float alpha_g[2] = {*p.g[0].epi.adam_alpha, p.tiles_total > p.tiles0 ? *p.g[1].epi.adam_alpha : 0.f};
  // TF ApplyAdam on quantum q of tile `id`, whose gradient is in cbuf.  A float4 that straddles column N is updated whole: the pad
  // columns of every tensor and of every operand are zero, and Adam on a zero gradient of a zero parameter is a fixed point.
  auto update = [&](auto qc, int id) {
    constexpr int q = decltype(qc)::value;
    int pi, tm, tn;
    decode(id, pi, tm, tn);
    const WgsProd& g = p.g[pi];
    const EpiD& e = g.epi;
    float* __restrict__ theta_out = e.adam_theta_out ? e.adam_theta_out : e.adam_theta;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int row_l = q * 32 + j * 16 + r16;
      const int row = tm * WGS_BM + row_l, col = tn * WGS_BN + 4 * c4;
      const bool ok = row < g.M && col < g.N;
      float4 v = *reinterpret_cast<const float4*>(cbuf + row_l * WGS_BN + 4 * c4);
      if (e.sp_rows != nullptr && ok) {
        const float4 x = sparse_rows_quad(e, row, col);
        v.x += x.x; v.y += x.y; v.z += x.z; v.w += x.w;
      }
      const size_t off = (size_t)row * g.ldc + col;
      adam_update(v.x, alpha, e.adam_reg, T[q][j].x, Mo[q][j].x, V[q][j].x, sq);
      adam_update(v.y, alpha, e.adam_reg, T[q][j].y, Mo[q][j].y, V[q][j].y, sq);
      adam_update(v.z, alpha, e.adam_reg, T[q][j].z, Mo[q][j].z, V[q][j].z, sq);
      adam_update(v.w, alpha, e.adam_reg, T[q][j].w, Mo[q][j].w, V[q][j].w, sq);
      *reinterpret_cast<float4*>(ok ? theta_out + off : dumpf) = T[q][j];
      *reinterpret_cast<float4*>(ok ? e.adam_m + off : dumpf) = Mo[q][j];
      *reinterpret_cast<float4*>(ok ? e.adam_v + off : dumpf) = V[q][j];
    }
  };
  if (ntile == 0) return;
  static_for<0, 2>([&](auto qc) { request(qc, wgs_tile_of(p.table, wg, nwg, 0)); });
  WGS_STAMP(1, 0);
  // Iteration r = 0 .. ntile: the GEMM waves multiply tile r (r < ntile) while these waves update tile r - 1 (r > 0) out of ring slot
  // q = 0 .. 3 and, as each slot is freed, request tile r's theta / m / v into it -- a whole iteration before their use in r + 1.
  // The four quanta are spread evenly over the stage barriers of the iteration (requests and stores leave at a steady rate), and
  // every iteration is straight-line code as far as memory operations go (first / middle / last iteration are instantiated apart),
  // so that hipcc's counted vmcnt waits leave the younger requests in flight.
  auto iteration = [&](int r, auto upd_c, auto req_c) {
    constexpr bool UPD = decltype(upd_c)::value, REQ = decltype(req_c)::value;
    const int id_gemm = r < ntile ? wgs_tile_of(p.table, wg, nwg, r) : -1;
    const int id_prev = UPD ? wgs_tile_of(p.table, wg, nwg, r - 1) : -1;
    int nst = 0;
    if (id_gemm >= 0) { int pi, tm, tn; decode(id_gemm, pi, tm, tn); nst = (p.g[pi].K + WGS_BK - 1) / WGS_BK; }
    if constexpr (UPD) { alpha = id_prev >= p.tiles0 ? alpha_g[1] : alpha_g[0]; sq = 0.f; }
    static_for<0, 2>([&](auto qc) {
      constexpr int q = decltype(qc)::value;
      const int s0 = (q * nst) >> 1, s1 = ((q + 1) * nst) >> 1;      // this quantum's share of the stage barriers
      if (s1 > s0) __builtin_amdgcn_s_barrier();
      if (r == 2) WGS_STAMP(1, 19 + 2 * q);
#ifdef GANMF_PERSIST_DIAG_BUILD
      if (!(p.diag & 1))
#endif
      {
      if constexpr (UPD) update(qc, id_prev);
      if constexpr (UPD && REQ) request(qc, id_gemm);
      }
      if (r == 2) WGS_STAMP(1, 20 + 2 * q);
      for (int s = s0 + 1; s < s1; ++s) __builtin_amdgcn_s_barrier();
    });
    float* sq_out = nullptr;
    if constexpr (UPD) {
      int pi, tm, tn;
      decode(id_prev, pi, tm, tn);
      const WgsProd& g = p.g[pi];
      if (g.epi.sq_partials) sq_out = g.epi.sq_partials + (size_t)tn * g.tiles_m + tm;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
      if (lane == 0) sqbuf[aw] = sq;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    WGS_STAMP(1, 1 + 3 * r);
    __builtin_amdgcn_s_barrier();      // Y: done with the previous gradient tile (the GEMM waves overwrite it now)
    WGS_STAMP(1, 2 + 3 * r);
    if (sq_out && at == 0) {
      float t = (sqbuf[0] + sqbuf[1]) + (sqbuf[2] + sqbuf[3]);
      t += (sqbuf[4] + sqbuf[5]) + (sqbuf[6] + sqbuf[7]);
      *sq_out = t;
    }
    __builtin_amdgcn_s_barrier();      // X: tile r's gradient is in LDS
    WGS_STAMP(1, 3 + 3 * r);
  };
  iteration(0, std::false_type{}, std::false_type{});
  for (int r = 1; r < ntile; ++r) iteration(r, std::true_type{}, std::true_type{});
  iteration(ntile, std::true_type{}, std::false_type{});
}

__global__ __launch_bounds__(256) void presplit_red_kernel(const RedP r, const RedPlanes rp, const int nred, const PlaneJobs js) {
  __shared__ __attribute__((aligned(16))) float red[4 + 4 * 256];
  if ((int)blockIdx.x < nred) {
    RedP q = r;
    if (rp.on) { q.epi.planes = rp.pl.p; q.epi.plane_stride = rp.pl.pstride; }
    splitk_reduce_body(q, (int)blockIdx.x, nred, 0, red);
  } else presplit_body(js, (int)blockIdx.x - nred, (int)gridDim.x - nred);
}

// Host: the schedule for two products of tiles_m[i] x tiles_n[i] tiles (tile id = product offset + tm * tiles_n + tn) on nwg
// workgroups (a multiple of 8).  Blocks of 4 x 8 tiles are dealt out to the XCDs in turn; a block's tiles go to the XCD's
// workgroups li = 0 .. nwg / 8 - 1 (workgroup li * 8 + x), each workgroup's list is kept dense.  Returns the number of rounds.
inline int wgs_build_schedule(const int tiles_m[2], const int tiles_n[2], int nprod, int nwg, std::vector<int>& table) {
  const int per_x = nwg / 8;
  std::vector<std::vector<int>> lists((size_t)nwg);
  int block = 0, base = 0;
  for (int pi = 0; pi < nprod; ++pi) {
    const int BMB = 4, BNB = 8;
    for (int bm = 0; bm < (tiles_m[pi] + BMB - 1) / BMB; ++bm)
      for (int bn = 0; bn < (tiles_n[pi] + BNB - 1) / BNB; ++bn, ++block) {
        const int x = block & 7;
        // the block's tiles to the workgroups of XCD x that have the fewest tiles so far (dense lists, balanced inside the XCD)
        std::vector<int> ids;
        for (int i = 0; i < BMB; ++i)
          for (int j = 0; j < BNB; ++j) {
            const int tm = bm * BMB + i, tn = bn * BNB + j;
            if (tm < tiles_m[pi] && tn < tiles_n[pi]) ids.push_back(base + tm * tiles_n[pi] + tn);
          }
        for (size_t t = 0; t < ids.size(); ++t) {
          int best = 0;
          for (int li = 1; li < per_x; ++li)
            if (lists[(size_t)li * 8 + x].size() < lists[(size_t)best * 8 + x].size()) best = li;
          lists[(size_t)best * 8 + x].push_back(ids[t]);
        }
      }
    base += tiles_m[pi] * tiles_n[pi];
  }
  size_t rounds = 0;
  for (auto& l : lists) rounds = std::max(rounds, l.size());
  table.assign(rounds * (size_t)nwg, -1);
  for (int w = 0; w < nwg; ++w)
    for (size_t i = 0; i < lists[(size_t)w].size(); ++i) table[i * (size_t)nwg + w] = lists[(size_t)w][i];
  return (int)rounds;
}

}  // namespace ganmf
