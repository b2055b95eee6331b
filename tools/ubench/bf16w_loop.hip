// Prototype of a 16-wave split-bf16 GEMM whose BOTH operands arrive already split into three bf16 planes (tile-major,
// pre-swizzled: a 64-row x 32-k block of one piece is 4 KiB, stored exactly as it sits in LDS), so that the K loop is
// LDS-DMA + fragment reads + MFMA only.  Question it answers: how long does a launch of the step's small-M shapes take in
// that form (the fp32-MFMA ring kernel: 21.3 us for 256 x 992 x 3707 in 4 slices, 15 us for 128 x 3706 x 993 in 2)?
//   workgroup: 1024 threads = 4 K groups x (2 x 2 waves), tile 128 x 64, K-tile 32, ring of 4 (36 KiB each);
//   K-tile t is computed by groups 2 (t & 1) and 2 (t & 1) + 1 (one 16-wide chunk each), 12 MFMAs per wave and tile;
//   the four groups' sums meet in LDS, one slab per K slice is written.
// hipcc --offload-arch=gfx950 -O3 bf16w_loop.hip -o bf16w_loop
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int BM = 128, BN = 64, BK = 32, NS = 4;
constexpr int BLK = 1024;                          // dwords of one 64-row x 32-k block of one piece
constexpr int A_DW = 2 * 3 * BLK, B_DW = 3 * BLK;  // per K-tile: A = two row blocks x three pieces, B = one x three
constexpr int STAGE = A_DW + B_DW;                 // 9216 dwords = 36 KiB
constexpr int LINES = STAGE / 4;                   // 16-byte lines per K-tile: 2304 = 2 x 1024 + 256

struct P {
  const unsigned* A;   // planes [M / 64][nkt][3][BLK]
  const unsigned* B;   // planes [N / 64][nkt][3][BLK]
  float* C;            // slabs [nsplit][M][N]
  int M, N, nkt, tiles_m, tiles_n, nsplit, kt_per_split;
};

__device__ __forceinline__ void glds16(const unsigned* src, unsigned* lds_wave_base) {
  const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void*)lds_wave_base);
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
}

template <int W3>      // 1: this wave issues three pieces per K-tile (waves 0-3), 0: two
__device__ __forceinline__ void body(const P& p, unsigned* smem) {
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kg = wave >> 2, wr = (wave >> 1) & 1, wc = wave & 1, li = lane & 31, lh = lane >> 5;
  const int bid = blockIdx.x;
  const int tm = bid % p.tiles_m, tn = (bid / p.tiles_m) % p.tiles_n, sp = bid / (p.tiles_m * p.tiles_n);
  const int kt0 = sp * p.kt_per_split, kt1 = min(p.nkt, kt0 + p.kt_per_split), nt = kt1 - kt0;
  // line l of a K-tile's stage: [0, 768) A row block 0, [768, 1536) A row block 1, [1536, 2304) B; each a contiguous 12 KiB in global
  const unsigned* src[3];
  int dstl[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int l = j * 1024 + tid;      // (j == 2: only tid < 256, i.e. waves 0-3)
    const int seg = l / 768, off = l % 768;
    const unsigned* base = seg < 2 ? p.A + ((size_t)(tm * 2 + seg) * p.nkt + kt0) * 3 * BLK : p.B + ((size_t)tn * p.nkt + kt0) * 3 * BLK;
    src[j] = base + off * 4;
    dstl[j] = (j * 1024 + wave * 64) * 4;      // dword offset of this wave's 1 KiB inside the stage
  }
  auto issue = [&](int slot) {
    unsigned* s = smem + slot * STAGE;
    glds16(src[0], s + dstl[0]); src[0] += 3 * BLK;
    glds16(src[1], s + dstl[1]); src[1] += 3 * BLK;
    if (W3) { glds16(src[2], s + dstl[2]); src[2] += 3 * BLK; }
  };
  f32x16 acc[2], accl[2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[a][r] = 0.f; accl[a][r] = 0.f; }

  constexpr int PW = W3 ? 3 : 2;      // pieces this wave has in flight per K-tile
#pragma unroll
  for (int s = 0; s < NS - 1; ++s)
    if (s < nt) issue(s);
  for (int t = 0; t < nt; ++t) {
    // tile t has landed when at most the younger tiles t+1, t+2 are outstanding
    const int younger = min(nt - 1 - t, NS - 2);
    if (younger >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PW) : "memory");
    else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PW) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (t + NS - 1 < nt) issue((t + NS - 1) % NS);      // slot of tile t-1: every wave is past its reads
#ifndef BF16W_STREAM_ONLY
    if ((kg >> 1) == (t & 1)) {
      const unsigned* s = smem + (t % NS) * STAGE;
      const int c = kg & 1;      // 16-wide chunk of the K-tile
      u32x4 fa[2][3], fb[3];
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          const int row = a * 32 + li;      // inside row block wr
          fa[a][q] = *reinterpret_cast<const u32x4*>(s + (wr * 3 + q) * BLK + row * 16 + 4 * ((2 * c + lh) ^ ((row >> 2) & 3)));
        }
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const int row = wc * 32 + li;
        fb[q] = *reinterpret_cast<const u32x4*>(s + A_DW + q * BLK + row * 16 + 4 * ((2 * c + lh) ^ ((row >> 2) & 3)));
      }
      constexpr int ta[6] = {1, 0, 2, 1, 0, 0}, tb[6] = {1, 2, 0, 0, 1, 0};
#pragma unroll
      for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int a = 0; a < 2; ++a) {
          const bf16x8 x = __builtin_bit_cast(bf16x8, fa[a][ta[i]]), y = __builtin_bit_cast(bf16x8, fb[tb[i]]);
          if (i < 5) accl[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, accl[a], 0, 0, 0);
          else acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc[a], 0, 0, 0);
        }
    }
#endif      // (-DBF16W_STREAM_ONLY: the LDS-DMA ring alone, no fragment reads, no MFMAs: what the operand streaming costs by itself)
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  // the four groups' sums meet in LDS: [kg][128][64] floats = 128 KiB of the 144 KiB ring
  float* red = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = wr * 64 + a * 32 + (r >> 2) * 8 + lh * 4 + (r & 3), col = wc * 32 + li;      // 32x32 accumulator layout
      red[(kg * BM + row) * BN + (col ^ ((row & 7) << 2))] = acc[a][r] + accl[a][r];
    }
  __syncthreads();
  float* C = p.C + (size_t)sp * p.M * p.N;
#pragma unroll
  for (int j = 0; j < BM * BN / 4 / 1024; ++j) {
    const int idx = j * 1024 + tid, row = idx / (BN / 4), c4 = idx % (BN / 4);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 v = *reinterpret_cast<const float4*>(red + (g * BM + row) * BN + ((c4 * 4) ^ ((row & 7) << 2)));
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    *reinterpret_cast<float4*>(C + (size_t)(tm * BM + row) * p.N + tn * BN + c4 * 4) = s;
  }
}

__global__ __launch_bounds__(1024) void bf16w_kernel(const P p) {
  extern __shared__ __attribute__((aligned(16))) unsigned smem[];
  if (threadIdx.x < 256) body<1>(p, smem);
  else body<0>(p, smem);
}

static unsigned short bf16_bits(float x) {
  unsigned u; memcpy(&u, &x, 4);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
static float bf16_val(unsigned short b) { unsigned u = (unsigned)b << 16; float f; memcpy(&f, &u, 4); return f; }

// planes of a [R][K] matrix (R % 64 == 0, K % 32 == 0): piece q of x = q-th term of the exact three-way split
static void make_planes(const std::vector<float>& x, int R, int K, std::vector<unsigned>& planes, std::vector<float>& rounded) {
  const int nkt = K / 32;
  planes.assign((size_t)(R / 64) * nkt * 3 * BLK, 0u);
  rounded.assign(x.size(), 0.f);
  unsigned short* h = reinterpret_cast<unsigned short*>(planes.data());
  for (int r = 0; r < R; ++r)
    for (int k = 0; k < K; ++k) {
      float rem = x[(size_t)r * K + k], sum = 0.f;
      for (int q = 0; q < 3; ++q) {
        const unsigned short b = bf16_bits(rem);
        const float v = bf16_val(b);
        rem -= v; sum += v;
        const int rt = r / 64, row = r % 64, kt = k / 32, kk = k % 32, chunk = kk / 8, e = kk % 8;
        const size_t dw = ((size_t)(rt * nkt + kt) * 3 + q) * BLK + row * 16 + 4 * (chunk ^ ((row >> 2) & 3)) + e / 2;
        h[dw * 2 + (e & 1)] = b;
      }
      rounded[(size_t)r * K + k] = sum;
    }
}

int main() {
  struct Shape { int M, N, K, nsplit; const char* what; };
  const Shape shapes[] = {{256, 1024, 3840, 8, "encode-like 256 x 992 x 3707, 8 K slices"},
                          {256, 1024, 3840, 4, "encode-like, 4 K slices (128 workgroups)"},
                          {128, 3712, 1024, 4, "decode-like 128 x 3706 x 993, 4 K slices"},
                          {128, 3712, 1024, 2, "decode-like, 2 K slices (116 workgroups)"},
                          {128, 1024, 3712, 16, "dE-like 128 x 992 x 3706, 16 K slices"},
                          {256, 3712, 1024, 2, "D-step decode-like 256 x 3706 x 993, 2 K slices"}};
  CHECK(hipFuncSetAttribute((const void*)bf16w_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, NS * STAGE * 4));
  hipStream_t st; CHECK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (const Shape& s : shapes) {
    std::vector<float> a((size_t)s.M * s.K), b((size_t)s.N * s.K), ar, br;
    srand(1);
    for (auto& v : a) v = (rand() % 2001 - 1000) / 1000.f;
    for (auto& v : b) v = (rand() % 2001 - 1000) / 3000.f;
    std::vector<unsigned> pa, pb;
    make_planes(a, s.M, s.K, pa, ar);
    make_planes(b, s.N, s.K, pb, br);
    unsigned *dA, *dB; float* dC;
    CHECK(hipMalloc(&dA, pa.size() * 4)); CHECK(hipMalloc(&dB, pb.size() * 4)); CHECK(hipMalloc(&dC, (size_t)s.nsplit * s.M * s.N * 4));
    CHECK(hipMemcpy(dA, pa.data(), pa.size() * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dB, pb.data(), pb.size() * 4, hipMemcpyHostToDevice));
    P p{dA, dB, dC, s.M, s.N, s.K / 32, s.M / BM, s.N / BN, s.nsplit, (s.K / 32 + s.nsplit - 1) / s.nsplit};
    const int grid = p.tiles_m * p.tiles_n * p.nsplit;
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
      CHECK(hipEventRecord(e0, st));
      for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(bf16w_kernel, dim3(grid), dim3(1024), NS * STAGE * 4, st, p);
      CHECK(hipEventRecord(e1, st)); CHECK(hipEventSynchronize(e1));
      CHECK(hipEventElapsedTime(&ms, e0, e1));
    }
    CHECK(hipGetLastError());
    std::vector<float> c((size_t)s.nsplit * s.M * s.N);
    CHECK(hipMemcpy(c.data(), dC, c.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0;
    for (int t = 0; t < 64; ++t) {
      const int m = (t * 37) % s.M, n = (t * 101 + 7) % s.N;
      double ref = 0, got = 0;
      for (int k = 0; k < s.K; ++k) ref += (double)a[(size_t)m * s.K + k] * b[(size_t)n * s.K + k];
      for (int q = 0; q < s.nsplit; ++q) got += c[((size_t)q * s.M + m) * s.N + n];
      worst = std::max(worst, std::fabs(got - ref));
    }
    const double us = ms * 1e3 / 200, fl = 2.0 * s.M * s.N * s.K;
    printf("%-52s %4d workgroups  %6.2f us per launch  %6.1f TFLOP/s fp32-equivalent  max |err| %.2e (|C| ~ %.1f)\n", s.what, grid, us, fl / us / 1e6, worst,
           std::sqrt((double)s.K) * 0.19);
    (void)hipFree(dA); (void)hipFree(dB); (void)hipFree(dC);
  }
  return 0;
}
