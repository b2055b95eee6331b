// Micro-benchmark: how fast can 89.5 MB be written as 128x128 fp32 tiles (row stride ld) vs linearly?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ __launch_bounds__(256) void tile_store(float* C, int ld, int tiles_m, int M, int N, int mode) {
  const int tm = blockIdx.x % tiles_m, tn = blockIdx.x / tiles_m;
  const int tid = threadIdx.x, tc = tid % 32, tr = tid / 32;
  const int col = tn * 128 + tc * 4;
  float4 v = make_float4(tid, tm, tn, 1.f);
  for (int j = 0; j < 16; ++j) {
    const int row = tm * 128 + tr + j * 8;
    if (row < M && col + 3 < N) {
      float4* p = reinterpret_cast<float4*>(C + (size_t)row * ld + col);
      typedef float f4 __attribute__((ext_vector_type(4)));
      f4 w = {v.x, v.y, v.z, v.w};
      if (mode == 1) __builtin_nontemporal_store(w, reinterpret_cast<f4*>(p)); else *p = v;
    }
  }
}
__global__ __launch_bounds__(256) void linear_store(float4* C, size_t n4) {
  float4 v = make_float4(1, 2, 3, 4);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) C[i] = v;
}
// tile store but each workgroup owns a CONTIGUOUS 64 KiB chunk (same bytes, different placement)
__global__ __launch_bounds__(256) void chunk_store(float4* C) {
  float4 v = make_float4(1, 2, 3, 4);
  float4* p = C + (size_t)blockIdx.x * 4096;
  for (int j = 0; j < 16; ++j) p[j * 256 + threadIdx.x] = v;
}
int main() {
  const int M = 6040, N = 3706, ld = 3712, tm = 48, tn = 29;
  float* C; CK(hipMalloc(&C, (size_t)6144 * ld * 4 + 4096));
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  auto time = [&](auto fn, const char* name, double bytes) {
    fn(); hipDeviceSynchronize();
    hipEventRecord(a); for (int i = 0; i < 20; ++i) fn(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("%-28s %8.2f us  %6.2f TB/s\n", name, ms / 20 * 1e3, bytes / (ms / 20 * 1e-3) / 1e12);
  };
  const double bytes = (double)M * N * 4;
  time([&] { hipLaunchKernelGGL(tile_store, dim3(tm * tn), dim3(256), 0, 0, C, ld, tm, M, N, 0); }, "tile 128x128 plain", bytes);
  time([&] { hipLaunchKernelGGL(tile_store, dim3(tm * tn), dim3(256), 0, 0, C, ld, tm, M, N, 1); }, "tile 128x128 nontemporal", bytes);
  time([&] { hipLaunchKernelGGL(chunk_store, dim3(tm * tn), dim3(256), 0, 0, (float4*)C); }, "64KiB contiguous per WG", 1392.0 * 65536);
  time([&] { hipLaunchKernelGGL(linear_store, dim3(2048), dim3(256), 0, 0, (float4*)C, (size_t)M * ld / 4); }, "linear grid-stride", (double)M * ld * 4);
  return 0;
}
