// What does a global store cost the fp32 MFMA stream of its SIMD?  512-thread workgroups, one per CU: waves 0-3 (one per
// SIMD) run ROUNDS of 16 x v_mfma_f32_32x32x2_f32 (1024 MFMA cycles, two dependent chains); waves 4-7 (their SIMD
// partners) do, per round of the MFMA waves' time, nothing / S x global_store_dwordx4 (1 KiB per wave-instruction, to an
// L2-resident 64 KiB window per workgroup) / S x global_store_dword / S x ds_read_b128 + global_store_dwordx4.
// A further mode lets the MFMA waves issue the stores themselves between their MFMAs.
// Reports cycles per round of the MFMA waves.  Build: hipcc -O3 --offload-arch=gfx950 mfma_store.hip -o mfma_store
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define MF(q) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc[(q) & 1]) : "v"(a), "v"(b))

template <int MODE, int S>
__global__ __launch_bounds__(512) void k(const float* in, float* out, float* win, long long* cyc, int iters) {
  __shared__ __attribute__((aligned(16))) float lds[16384];
  const int tid = threadIdx.x, wave = tid >> 6;
  f32x16 acc[2] = {};
  const float a = in[tid], b = in[tid + 1];
  f32x4 v = {in[tid + 2], in[tid + 3], in[tid + 4], in[tid + 5]};
  lds[tid] = 0.f;
  __syncthreads();
  float* dst = win + (size_t)blockIdx.x * 16384 + (tid & 255) * 4;   // 1 KiB per wave-instruction, 4 KiB per round of 4 waves
  const unsigned lds_addr = (unsigned)((tid & 255) * 16);
  const long long t0 = __builtin_readcyclecounter();
  if (wave < 4) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        MF(q >> 3);
        if (MODE == 4 && q < S) {     // the MFMA wave stores by itself
          asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(dst + (size_t)((it * S + q) & 15) * 1024), "v"(v) : "memory");
        }
      }
    }
  } else {
    for (int it = 0; it < iters; ++it) {
      if (MODE == 1) {
#pragma unroll
        for (int j = 0; j < S; ++j)
          asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(dst + (size_t)((it * S + j) & 15) * 1024), "v"(v) : "memory");
      } else if (MODE == 2) {
#pragma unroll
        for (int j = 0; j < S; ++j)
          asm volatile("global_store_dword %0, %1, off" ::"v"(dst + (size_t)((it * S + j) & 15) * 1024), "v"(v[0]) : "memory");
      } else if (MODE == 3) {
#pragma unroll
        for (int j = 0; j < S; ++j) {
          f32x4 r;
          asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(lds_addr) : "memory");
          asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(dst + (size_t)((it * S + j) & 15) * 1024), "v"(r) : "memory");
        }
      }
      // pace: about one round of the MFMA waves (1024 cycles) per iteration
      if (MODE != 4) __builtin_amdgcn_s_sleep(14);
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = v[0];
  for (int q = 0; q < 2; ++q) for (int r = 0; r < 16; ++r) s += acc[q][r];
  out[blockIdx.x * 512 + tid] = s;
  if ((tid & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int MODE, int S>
void run(const char* name, const float* in, float* out, float* win, long long* cyc) {
  const int iters = 2000, wgs = 256;
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL((k<MODE, S>), dim3(wgs), dim3(512), 0, 0, in, out, win, cyc, iters);
    hipDeviceSynchronize();
  }
  long long h[256 * 8];
  hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
  double mf = 0, st = 0;
  for (int i = 0; i < wgs; ++i) { for (int w = 0; w < 4; ++w) mf += h[i * 8 + w]; for (int w = 4; w < 8; ++w) st += h[i * 8 + w]; }
  printf("%-58s MFMA waves %7.1f cycles per 16-MFMA round (ideal 1024); partner waves %7.1f per iteration\n", name,
         mf / (wgs * 4) / iters, st / (wgs * 4) / iters);
}

int main() {
  float *in, *out, *win; long long* cyc;
  hipMalloc(&in, 8192); hipMemset(in, 0, 8192);
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&win, (size_t)256 * 16384 * 4); hipMalloc(&cyc, 256 * 8 * 8);
  run<0, 0>("partner idle (sleeping)", in, out, win, cyc);
  run<1, 1>("partner: 1 x global_store_dwordx4 per round", in, out, win, cyc);
  run<1, 2>("partner: 2 x global_store_dwordx4 per round", in, out, win, cyc);
  run<1, 4>("partner: 4 x global_store_dwordx4 per round", in, out, win, cyc);
  run<2, 4>("partner: 4 x global_store_dword per round", in, out, win, cyc);
  run<3, 2>("partner: 2 x (ds_read_b128 + global_store_dwordx4) per round", in, out, win, cyc);
  run<4, 1>("MFMA wave itself: 1 x global_store_dwordx4 per round", in, out, win, cyc);
  run<4, 2>("MFMA wave itself: 2 x global_store_dwordx4 per round", in, out, win, cyc);
  run<4, 4>("MFMA wave itself: 4 x global_store_dwordx4 per round", in, out, win, cyc);
  return 0;
}
