// Which SIMD does wave w of a workgroup run on?  (gfx950: s_getreg_b32 HW_REG_HW_ID -- wave_id [3:0], simd_id [5:4], cu_id [11:8], se_id [15:13])
// build: hipcc --offload-arch=gfx950 -O2 tools/ubench/wave_simd.hip -o tools/ubench/wave_simd ; run: tools/ubench/wave_simd [threads per block]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ void probe(unsigned* out) {
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    out[(blockIdx.x * (blockDim.x >> 6) + wave) * 2] = id;
    out[(blockIdx.x * (blockDim.x >> 6) + wave) * 2 + 1] = xcc;
  }
}
int main(int argc, char** argv) {
  const int threads = argc > 1 ? atoi(argv[1]) : 1024, blocks = 8, waves = threads / 64;
  unsigned* d;
  hipMalloc(&d, blocks * waves * 2 * sizeof(unsigned));
  hipLaunchKernelGGL(probe, dim3(blocks), dim3(threads), 0, 0, d);
  std::vector<unsigned> h(blocks * waves * 2);
  hipMemcpy(h.data(), d, h.size() * sizeof(unsigned), hipMemcpyDeviceToHost);
  for (int b = 0; b < blocks; ++b) {
    printf("block %d (xcc %u, cu %u):", b, h[b * waves * 2 + 1] & 15, (h[b * waves * 2] >> 8) & 15);
    for (int w = 0; w < waves; ++w) printf(" w%d->simd%u", w, (h[(b * waves + w) * 2] >> 4) & 3);
    printf("\n");
  }
  return 0;
}
