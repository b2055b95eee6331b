// Does VALU work overlap with bf16 MFMAs on one SIMD?  Each wave runs ITER rounds of
//   NM x v_mfma_f32_32x32x16_bf16 (4 independent accumulators)  +  NV x VALU ops (and / pk_add / perm mix)
// in three orders: MFMAs only, VALU only, blocked (all MFMAs then all VALU), fine (1 MFMA : NV/NM VALU).
// Launched with 1 and 2 waves per SIMD.  Build: hipcc -O3 --offload-arch=gfx950 mfma_valu.hip -o mfma_valu
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void k(const unsigned* in, float* out, long long* cyc, int iters) {
  f32x16 acc[4] = {};
  u32x4 a = {in[threadIdx.x], in[threadIdx.x + 1], in[threadIdx.x + 2], in[threadIdx.x + 3]};
  u32x4 b = {in[threadIdx.x + 4], in[threadIdx.x + 5], in[threadIdx.x + 6], in[threadIdx.x + 7]};
  unsigned v[8];
  for (int j = 0; j < 8; ++j) v[j] = in[threadIdx.x + 8 + j];
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#define MF(q) acc[q & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc[q & 3], 0, 0, 0)
#define VA(j) { v[j & 7] = (v[j & 7] & 0xffff0000u) + v[(j + 1) & 7]; v[(j + 3) & 7] = __builtin_amdgcn_perm(v[j & 7], v[(j + 5) & 7], 0x07060302u); \
                v[(j + 2) & 7] = __float_as_uint(__uint_as_float(v[(j + 2) & 7]) - __uint_as_float(v[(j + 6) & 7])); }
    if (MODE == 0) {
#pragma unroll
      for (int q = 0; q < 24; ++q) MF(q);
    } else if (MODE == 1) {
#pragma unroll
      for (int j = 0; j < 48; ++j) VA(j);     // 48 x 4 = 192 VALU ops
    } else if (MODE == 2) {
#pragma unroll
      for (int q = 0; q < 24; ++q) MF(q);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 48; ++j) VA(j);
      __builtin_amdgcn_sched_barrier(0);
    } else {
#pragma unroll
      for (int q = 0; q < 24; ++q) {
        MF(q);
        __builtin_amdgcn_sched_barrier(0);
        VA(2 * q); VA(2 * q + 1);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int q = 0; q < 4; ++q) for (int r = 0; r < 16; ++r) s += acc[q][r];
  for (int j = 0; j < 8; ++j) s += __uint_as_float(v[j]);
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
  unsigned* in; float* out; long long* cyc;
  hipMalloc(&in, 4096); hipMemset(in, 0, 4096);
  hipMalloc(&out, 1024 * 256 * 4); hipMalloc(&cyc, 1024 * 8);
  const int iters = 2000;
  const char* names[4] = {"24 MFMA only", "192 VALU only", "blocked 24 MFMA + 192 VALU", "fine 1 MFMA : 8 VALU"};
  for (int wgs = 256; wgs <= 512; wgs *= 2) {     // 256 = 1 wave/SIMD, 512 = 2 waves/SIMD
    for (int m = 0; m < 4; ++m) {
      for (int rep = 0; rep < 2; ++rep) {
        if (m == 0) hipLaunchKernelGGL(k<0>, dim3(wgs), dim3(256), 0, 0, in, out, cyc, iters);
        if (m == 1) hipLaunchKernelGGL(k<1>, dim3(wgs), dim3(256), 0, 0, in, out, cyc, iters);
        if (m == 2) hipLaunchKernelGGL(k<2>, dim3(wgs), dim3(256), 0, 0, in, out, cyc, iters);
        if (m == 3) hipLaunchKernelGGL(k<3>, dim3(wgs), dim3(256), 0, 0, in, out, cyc, iters);
        hipDeviceSynchronize();
      }
      long long h[1024]; hipMemcpy(h, cyc, wgs * 8, hipMemcpyDeviceToHost);
      double avg = 0; for (int i = 0; i < wgs; ++i) avg += h[i]; avg /= wgs;
      printf("%d waves/SIMD  %-30s %8.1f clk-counter ticks per round per wave\n", wgs / 256, names[m], avg / iters);
    }
  }
  return 0;
}
