// Micro-benchmark: what does one "K-tile" of the GEMM main loop cost, component by component?
// 256 workgroups x 256 threads (one wave per SIMD), 64 MFMAs (32x32x2 f32) per wave per tile.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE, int NACC>   // MODE bit0: ds_read fragments, bit1: barrier per tile, bit2: 8 glds per tile
__global__ __launch_bounds__(256) void loop(const float* __restrict__ src, float* out, unsigned long long* cyc, int ntiles) {
  __shared__ __attribute__((aligned(16))) float smem[2 * 8192];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  f32x16 acc[NACC];
  for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  for (int i = tid; i < 2 * 8192; i += 256) smem[i] = (float)(i & 7);
  __syncthreads();
  float4 fa = make_float4(1, 2, 3, 4), fb = make_float4(4, 3, 2, 1);
  const float* g = src + (size_t)blockIdx.x * 8192 + tid * 4;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int t = 0; t < ntiles; ++t) {
    float* buf = smem + (t & 1) * 8192;
    float4 stage[8];
    if (MODE & 8) {   // register staging: plain 16-byte loads now, ds_write_b128 after the MFMAs
#pragma unroll
      for (int j = 0; j < 8; ++j) stage[j] = *reinterpret_cast<const float4*>(g + j * 1024);
    }
    if (MODE & 4) {
#pragma unroll
      for (int j = 0; j < 8; ++j)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + j * 1024),
                                         (__attribute__((address_space(3))) void*)(buf + (j * 256 + wave * 64) * 4), 16, 0, 0);
    }
#pragma unroll
    for (int c = 0; c < 64 / (4 * NACC); ++c) {
      if (MODE & 1) {
        fa = *reinterpret_cast<const float4*>(smem + ((lane & 31) * 64 + 4 * ((2 * c + (lane >> 5)) ^ (lane & 15))) % 8192);
        fb = *reinterpret_cast<const float4*>(smem + 4096 + ((lane & 31) * 64 + 4 * ((2 * c + (lane >> 5)) ^ (lane & 15))) % 4096);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int a = 0; a < NACC; ++a) {
        acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.x, fb.x, acc[a], 0, 0, 0);
        acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.y, fb.y, acc[a], 0, 0, 0);
        acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.z, fb.z, acc[a], 0, 0, 0);
        acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.w, fb.w, acc[a], 0, 0, 0);
      }
    }
    if (MODE & 8) {
#pragma unroll
      for (int j = 0; j < 8; ++j) *reinterpret_cast<float4*>(buf + (j * 256 + tid) * 4) = stage[j];
    }
    if (MODE & 2) {
      if (MODE & 4) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
  float s = 0;
  for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
  out[blockIdx.x * 256 + tid] = s;
}

// producer/consumer split: waves 0-3 issue MFMAs (+ fragment reads), waves 4-7 only issue the glds (8 each
// per tile = the same 32 KiB) and wait for them; one barrier per tile joins all eight.
template <int NACC, int LOADER_WAVES>
__global__ __launch_bounds__(512) void loop_pc(const float* __restrict__ src, float* out, unsigned long long* cyc, int ntiles) {
  __shared__ __attribute__((aligned(16))) float smem[2 * 8192];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  f32x16 acc[NACC];
  for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  for (int i = tid; i < 2 * 8192; i += 512) smem[i] = (float)(i & 7);
  __syncthreads();
  float4 fa = make_float4(1, 2, 3, 4), fb = make_float4(4, 3, 2, 1);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (wave < 4) {
    for (int t = 0; t < ntiles; ++t) {
#pragma unroll
      for (int c = 0; c < 64 / (4 * NACC); ++c) {
        fa = *reinterpret_cast<const float4*>(smem + ((lane & 31) * 64 + 4 * ((2 * c + (lane >> 5)) ^ (lane & 15))) % 8192);
        fb = *reinterpret_cast<const float4*>(smem + 4096 + ((lane & 31) * 64 + 4 * ((2 * c + (lane >> 5)) ^ (lane & 15))) % 4096);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int a = 0; a < NACC; ++a) {
          acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.x, fb.x, acc[a], 0, 0, 0);
          acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.y, fb.y, acc[a], 0, 0, 0);
          acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.z, fb.z, acc[a], 0, 0, 0);
          acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.w, fb.w, acc[a], 0, 0, 0);
        }
      }
      __builtin_amdgcn_s_barrier();
    }
  } else {
    const int lw = wave - 4;
    const float* g = src + (size_t)blockIdx.x * 8192 + lane * 4;
    for (int t = 0; t < ntiles; ++t) {
      float* buf = smem + (t & 1) * 8192;
      if (lw < LOADER_WAVES) {
#pragma unroll
        for (int j = 0; j < 32 / LOADER_WAVES; ++j) {
          const int piece = lw * (32 / LOADER_WAVES) + j;
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + piece * 256),
                                           (__attribute__((address_space(3))) void*)(buf + piece * 256), 16, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0 && wave < 4) cyc[blockIdx.x * 4 + wave] = t1 - t0;
  float s = 0;
  for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
  out[blockIdx.x * 256 + (tid & 255)] = s;
}

// bytes-per-tile sweep: PIECES glds (1 KiB each per wave) per tile per wave, issued one tile ahead
template <int PIECES>
__global__ __launch_bounds__(256) void loop_bytes(const float* __restrict__ src, float* out, unsigned long long* cyc, int ntiles) {
  __shared__ __attribute__((aligned(16))) float smem[2 * PIECES * 1024];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  f32x16 acc[4];
  for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  float4 fa = make_float4(1, 2, 3, 4), fb = make_float4(4, 3, 2, 1);
  const float* g = src + (size_t)blockIdx.x * 8192 + tid * 4;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int t = 0; t < ntiles; ++t) {
    float* buf = smem + (t & 1) * PIECES * 1024;
#pragma unroll
    for (int j = 0; j < PIECES; ++j)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + (j & 7) * 1024),
                                       (__attribute__((address_space(3))) void*)(buf + (j * 256 + wave * 64) * 4), 16, 0, 0);
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.x, fb.x, acc[a], 0, 0, 0);
        acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.y, fb.y, acc[a], 0, 0, 0);
        acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.z, fb.z, acc[a], 0, 0, 0);
        acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.w, fb.w, acc[a], 0, 0, 0);
      }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
  float s = 0;
  for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
  out[blockIdx.x * 256 + tid] = s + smem[tid];
}
template <int PIECES>
void run_bytes(const float* src, float* out, unsigned long long* cyc, int ntiles) {
  hipLaunchKernelGGL((loop_bytes<PIECES>), dim3(256), dim3(256), 0, 0, src, out, cyc, ntiles);
  hipLaunchKernelGGL((loop_bytes<PIECES>), dim3(256), dim3(256), 0, 0, src, out, cyc, ntiles);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(1024);
  hipMemcpy(h.data(), cyc, 8192, hipMemcpyDeviceToHost);
  std::sort(h.begin(), h.end());
  const double c = (double)h[512] / ntiles;
  printf("glds %2d KiB per tile per CU + 64 MFMAs/wave : %7.1f cycles per tile  (%.1f B/clk/CU)\n", PIECES * 4, c, PIECES * 4096.0 / c);
}

template <int NACC, int LW>
void run_pc(const char* name, const float* src, float* out, unsigned long long* cyc, int ntiles) {
  hipLaunchKernelGGL((loop_pc<NACC, LW>), dim3(256), dim3(512), 0, 0, src, out, cyc, ntiles);
  hipLaunchKernelGGL((loop_pc<NACC, LW>), dim3(256), dim3(512), 0, 0, src, out, cyc, ntiles);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(1024);
  hipMemcpy(h.data(), cyc, 8192, hipMemcpyDeviceToHost);
  std::sort(h.begin(), h.end());
  printf("%-44s NACC=%d loaders=%d : %7.1f cycles per tile\n", name, NACC, LW, (double)h[512] / ntiles);
}

template <int MODE, int NACC>
void run(const char* name, const float* src, float* out, unsigned long long* cyc, int ntiles) {
  hipLaunchKernelGGL((loop<MODE, NACC>), dim3(256), dim3(256), 0, 0, src, out, cyc, ntiles);
  hipLaunchKernelGGL((loop<MODE, NACC>), dim3(256), dim3(256), 0, 0, src, out, cyc, ntiles);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(1024);
  hipMemcpy(h.data(), cyc, 8192, hipMemcpyDeviceToHost);
  std::sort(h.begin(), h.end());
  printf("%-44s NACC=%d : %7.1f cycles per tile (64 MFMAs = 4096 ideal)\n", name, NACC, (double)h[512] / ntiles);
}

int main() {
  float *src, *out; unsigned long long* cyc;
  hipMalloc(&src, 256 * 8192 * 4 + 65536); hipMemset(src, 0, 256 * 8192 * 4 + 65536);
  hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 8192);
  const int nt = 64;
  run<0, 4>("MFMA only", src, out, cyc, nt);
  run<0, 1>("MFMA only", src, out, cyc, nt);
  run<1, 4>("MFMA + ds_read_b128 frags", src, out, cyc, nt);
  run<1, 1>("MFMA + ds_read_b128 frags", src, out, cyc, nt);
  run<2, 4>("MFMA + barrier/tile", src, out, cyc, nt);
  run<3, 4>("MFMA + frags + barrier", src, out, cyc, nt);
  run<3, 1>("MFMA + frags + barrier", src, out, cyc, nt);
  run<6, 4>("MFMA + 8 glds + vmcnt0 + barrier", src, out, cyc, nt);
  run<7, 4>("MFMA + frags + 8 glds + vmcnt0 + barrier", src, out, cyc, nt);
  run<7, 1>("MFMA + frags + 8 glds + vmcnt0 + barrier", src, out, cyc, nt);
  run<4, 4>("MFMA + 8 glds (no wait)", src, out, cyc, nt);
  run<8 | 2, 4>("MFMA + 8 global_load + 8 ds_write + barrier", src, out, cyc, nt);
  run<8 | 2 | 1, 4>("MFMA + frags + regstage + barrier", src, out, cyc, nt);
  run<8 | 2 | 1, 1>("MFMA + frags + regstage + barrier", src, out, cyc, nt);
  run_bytes<2>(src, out, cyc, nt);
  run_bytes<4>(src, out, cyc, nt);
  run_bytes<8>(src, out, cyc, nt);
  run_bytes<12>(src, out, cyc, nt);
  run_bytes<16>(src, out, cyc, nt);
  run_pc<4, 4>("producer/consumer, 32 KiB per tile", src, out, cyc, nt);
  run_pc<4, 2>("producer/consumer, 32 KiB per tile", src, out, cyc, nt);
  run_pc<4, 1>("producer/consumer, 32 KiB per tile", src, out, cyc, nt);
  run_pc<1, 4>("producer/consumer, 32 KiB per tile", src, out, cyc, nt);
  return 0;
}
