// Micro-benchmark: sustained LDS read bandwidth per CU for the two fragment-read shapes of the staged bf16 kernels:
//   b128  one ds_read_b128 per lane (K-contiguous operand planes: 1 KiB per wave and instruction)
//   b32   four ds_read_b32 per lane, conflict-free rows (K-major operand planes: the same 1 KiB in four instructions)
// W waves per CU (one workgroup of 64 W threads per CU), ITER rounds of 8 fragments each; cycles by s_memtime (100 MHz -> the
// clock ratio is printed with the result).  hipcc --offload-arch=gfx950 -O3 lds_read.hip -o lds_read
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ void rd(unsigned* out, unsigned long long* ticks, int iters) {
  __shared__ __attribute__((aligned(16))) unsigned smem[24576];      // 96 KiB
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 24576; i += blockDim.x) smem[i] = i;
  __syncthreads();
  u32x4 acc = {0, 0, 0, 0};
  const unsigned* base = smem + (wave & 7) * 2048;
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
    base = smem + ((wave + it) & 7) * 2048;      // (a different operand tile every round: nothing loop-invariant to hoist)
    asm volatile("" ::: "memory");
#pragma unroll
    for (int f = 0; f < 8; ++f) {
      if (MODE == 0) {      // [row][16 dwords], chunk swizzled by the row: lane (i, h) reads 16 B
        const int i = lane & 31, h = lane >> 5;
        const u32x4 v = *reinterpret_cast<const u32x4*>(base + ((f * 32 + i) & 127) * 16 + 4 * (((f & 1) * 2 + h) ^ ((i >> 2) & 3)));
        acc += v;
      } else {              // [k/2][64 rows]: four k-pair rows, lane reads one dword of each
        const int i = lane & 31, h = lane >> 5;
        const unsigned* q = base + ((f & 3) * 4 * 4 + 4 * h) * 64 + ((i + 32 * (f & 1)) ^ (h << 5));
        acc[0] += q[0]; acc[1] += q[64]; acc[2] += q[128]; acc[3] += q[192];
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
  if (lane == 0) ticks[blockIdx.x * 16 + wave] = t1 - t0;
  out[blockIdx.x * blockDim.x + tid] = acc[0] + acc[1] + acc[2] + acc[3];
}

int main() {
  unsigned* out; unsigned long long* ticks;
  hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&ticks, 256 * 16 * 8);
  const int iters = 4000;
  for (int mode = 0; mode < 2; ++mode)
    for (int waves : {4, 8, 16}) {
      hipMemset(ticks, 0, 256 * 16 * 8);
      hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(a);
        if (mode == 0) hipLaunchKernelGGL(rd<0>, dim3(256), dim3(64 * waves), 0, 0, out, ticks, iters);
        else hipLaunchKernelGGL(rd<1>, dim3(256), dim3(64 * waves), 0, 0, out, ticks, iters);
        hipEventRecord(b); hipEventSynchronize(b);
      }
      float ms; hipEventElapsedTime(&ms, a, b);
      const double bytes_per_cu = (double)iters * 8 * 1024 * waves;      // 1 KiB per wave and fragment
      printf("%-5s %2d waves/CU: %8.1f us  %7.1f GB/s per CU  (= %.1f B/clk at 2.0 GHz, %.1f at 2.4 GHz)\n", mode ? "b32x4" : "b128", waves,
             ms * 1e3, bytes_per_cu / (ms * 1e-3) / 1e9, bytes_per_cu / (ms * 1e-3) / 2.0e9, bytes_per_cu / (ms * 1e-3) / 2.4e9);
    }
  return 0;
}
