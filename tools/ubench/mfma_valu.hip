// Does VALU work overlap with bf16 MFMAs on one SIMD?  Each wave runs ITER rounds of
//   24 x v_mfma_f32_32x32x16_bf16 (4 independent accumulators)  and / or  168 x VALU (v_and_b32 on 8 registers)
// in four orders: MFMAs only, VALU only, blocked (all MFMAs then all VALU), fine (1 MFMA : 7 VALU).
// Every instruction is its own `asm volatile`, so hipcc keeps the written order (a first version used builtins and
// sched_barrier: the compiler regrouped the MFMAs and the "fine" case was not fine at all).
// Launched with 1 and 2 waves per SIMD.  Build: hipcc -O3 --offload-arch=gfx950 mfma_valu.hip -o mfma_valu
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define MF(q) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[(q) & 3]) : "v"(a), "v"(b))
#define VA(j) asm volatile("v_and_b32 %0, 0xffff0fff, %0" : "+v"(v[(j) & 7]))
// the 11-instruction exact bf16 split of one element pair (gemm_bf16s.hpp), a dependent chain
#define SPLIT(j) asm volatile(                                                                     \
    "v_cvt_pk_bf16_f32 %0, %5, %6\n\tv_lshlrev_b32 %3, 16, %0\n\tv_and_b32 %4, 0xffff0000, %0\n\t"      \
    "v_sub_f32 %3, %5, %3\n\tv_sub_f32 %4, %6, %4\n\tv_cvt_pk_bf16_f32 %1, %3, %4\n\t"                  \
    "v_lshlrev_b32 %2, 16, %1\n\tv_sub_f32 %3, %3, %2\n\tv_and_b32 %2, 0xffff0000, %1\n\t"              \
    "v_sub_f32 %4, %4, %2\n\tv_cvt_pk_bf16_f32 %2, %3, %4"                                          \
    : "=&v"(sh), "=&v"(sm), "=&v"(sl), "=&v"(st0), "=&v"(st1) : "v"(fv[(j) & 7]), "v"(fv[((j) + 1) & 7]))

template <int MODE>
__global__ __launch_bounds__(256) void k(const unsigned* in, float* out, long long* cyc, int iters) {
  f32x16 acc[4] = {};
  u32x4 a = {in[threadIdx.x], in[threadIdx.x + 1], in[threadIdx.x + 2], in[threadIdx.x + 3]};
  u32x4 b = {in[threadIdx.x + 4], in[threadIdx.x + 5], in[threadIdx.x + 6], in[threadIdx.x + 7]};
  unsigned v[8];
  for (int j = 0; j < 8; ++j) v[j] = in[threadIdx.x + 8 + j];
  float fv[8];
  for (int j = 0; j < 8; ++j) fv[j] = __uint_as_float(in[threadIdx.x + 16 + j]);
  unsigned sh = 0, sm = 0, sl = 0; float st0, st1;
  __shared__ float ldsbuf[4096];
  ldsbuf[threadIdx.x] = 0.f;
  __syncthreads();
  u32x4 lr = {0, 0, 0, 0};
  const unsigned lds_addr = (unsigned)(threadIdx.x * 16);
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
#pragma unroll
      for (int q = 0; q < 24; ++q) MF(q);
    } else if (MODE == 1) {
#pragma unroll
      for (int j = 0; j < 168; ++j) VA(j);
    } else if (MODE == 2) {
#pragma unroll
      for (int q = 0; q < 24; ++q) MF(q);
#pragma unroll
      for (int j = 0; j < 168; ++j) VA(j);
    } else if (MODE == 3) {
#pragma unroll
      for (int q = 0; q < 24; ++q) {
        MF(q);
        VA(7 * q); VA(7 * q + 1); VA(7 * q + 2); VA(7 * q + 3); VA(7 * q + 4); VA(7 * q + 5); VA(7 * q + 6);
      }
    } else if (MODE == 4) {
#pragma unroll
      for (int q = 0; q < 12; ++q) SPLIT(q);
    } else if (MODE == 5) {
#pragma unroll
      for (int q = 0; q < 24; ++q) { MF(q); if (q & 1) SPLIT(q); }
    } else {
#pragma unroll
      for (int q = 0; q < 24; ++q) {
        MF(q);
        asm volatile("ds_read_b128 %0, %1" : "=v"(lr) : "v"(lds_addr) : "memory");
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int q = 0; q < 4; ++q) for (int r = 0; r < 16; ++r) s += acc[q][r];
  for (int j = 0; j < 8; ++j) s += __uint_as_float(v[j]);
  s += __uint_as_float(sh ^ sm ^ sl ^ lr[0] ^ lr[1] ^ lr[2] ^ lr[3]);
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
  unsigned* in; float* out; long long* cyc;
  hipMalloc(&in, 4096); hipMemset(in, 0, 4096);
  hipMalloc(&out, 1024 * 256 * 4); hipMalloc(&cyc, 1024 * 8);
  const int iters = 2000;
  const char* names[7] = {"24 MFMA only", "168 VALU only", "blocked 24 MFMA + 168 VALU", "fine 1 MFMA : 7 VALU",
                          "12 pair-splits only (132 VALU)", "24 MFMA, a pair-split every 2nd", "24 MFMA, a ds_read_b128 each"};
  for (int wgs = 256; wgs <= 512; wgs *= 2) {     // 256 = 1 wave/SIMD, 512 = 2 waves/SIMD
    for (int m = 0; m < 7; ++m) {
      for (int rep = 0; rep < 2; ++rep) {
        if (m == 0) hipLaunchKernelGGL(k<0>, dim3(wgs), dim3(256), 0, 0, in, out, cyc, iters);
        if (m == 1) hipLaunchKernelGGL(k<1>, dim3(wgs), dim3(256), 0, 0, in, out, cyc, iters);
        if (m == 2) hipLaunchKernelGGL(k<2>, dim3(wgs), dim3(256), 0, 0, in, out, cyc, iters);
        if (m == 3) hipLaunchKernelGGL(k<3>, dim3(wgs), dim3(256), 0, 0, in, out, cyc, iters);
        if (m == 4) hipLaunchKernelGGL(k<4>, dim3(wgs), dim3(256), 0, 0, in, out, cyc, iters);
        if (m == 5) hipLaunchKernelGGL(k<5>, dim3(wgs), dim3(256), 0, 0, in, out, cyc, iters);
        if (m == 6) hipLaunchKernelGGL(k<6>, dim3(wgs), dim3(256), 0, 0, in, out, cyc, iters);
        hipDeviceSynchronize();
      }
      long long h[1024]; hipMemcpy(h, cyc, wgs * 8, hipMemcpyDeviceToHost);
      double avg = 0; for (int i = 0; i < wgs; ++i) avg += h[i]; avg /= wgs;
      printf("%d waves/SIMD  %-30s %8.1f cycles per round per wave\n", wgs / 256, names[m], avg / iters);
    }
  }
  return 0;
}
