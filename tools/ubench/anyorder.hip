// Micro-benchmark: can two kernels of ONE stream overlap?  hipExtLaunchKernelGGL(..., flags = hipExtAnyOrderLaunch) clears the barrier bit of
// the kernel's AQL packet: the packet processor starts it without waiting for the packets in front of it.  Chain per iteration
//   A (ordered) -> B (ordered | any-order) -> C (ordered)
// with A and B spinning T us on a quarter of the chip each: ordered = 2T + C, overlapped = T + C.  Also the same overlap through a second
// stream (event fork after A's predecessor, join before C), which is what the data-parallel lanes of the library use.
// hipcc --offload-arch=gfx950 -O3 anyorder.hip -o anyorder
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void spin_kernel(unsigned long long ticks, int* out, int tag) {
  const unsigned long long t0 = wall_clock64();      // 100 MHz
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0) out[blockIdx.x] = tag;
}
// C checks that it runs behind BOTH: every slot of a and b carries this iteration's tag
__global__ void check_kernel(const int* a, const int* b, int n, int tag, int* bad) {
  for (int i = threadIdx.x; i < n; i += blockDim.x) if (a[i] != tag || b[i] != tag) atomicAdd(bad, 1);
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 200;
  const int us = argc > 2 ? atoi(argv[2]) : 20;
  const int WGS = 64;
  int *a, *b, *bad;
  CHECK(hipMalloc(&a, WGS * 4)); CHECK(hipMalloc(&b, WGS * 4)); CHECK(hipMalloc(&bad, 4));
  hipStream_t st, st2;
  CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking)); CHECK(hipStreamCreateWithFlags(&st2, hipStreamNonBlocking));
  hipEvent_t e0, e1, fork, join;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  CHECK(hipEventCreateWithFlags(&fork, hipEventDisableTiming)); CHECK(hipEventCreateWithFlags(&join, hipEventDisableTiming));
  const unsigned long long ticks = (unsigned long long)us * 100;
  const char* names[4] = {"ordered (flags 0)", "B any-order", "B on a second stream (event fork / join)", "B and C any-order (C must NOT be trusted)"};
  for (int mode = 0; mode < 4; ++mode) {
    float ms = 0;
    int hbad = 0;
    for (int rep = 0; rep < 3; ++rep) {
      CHECK(hipMemsetAsync(bad, 0, 4, st)); CHECK(hipMemsetAsync(a, 0, WGS * 4, st)); CHECK(hipMemsetAsync(b, 0, WGS * 4, st));
      CHECK(hipStreamSynchronize(st));
      CHECK(hipEventRecord(e0, st));
      for (int it = 1; it <= iters; ++it) {
        if (mode == 2) {
          CHECK(hipEventRecord(fork, st));
          CHECK(hipStreamWaitEvent(st2, fork, 0));
          hipLaunchKernelGGL(spin_kernel, dim3(WGS), dim3(256), 0, st, ticks, a, it);
          hipLaunchKernelGGL(spin_kernel, dim3(WGS), dim3(256), 0, st2, ticks, b, it);
          CHECK(hipEventRecord(join, st2));
          CHECK(hipStreamWaitEvent(st, join, 0));
          hipLaunchKernelGGL(check_kernel, dim3(1), dim3(64), 0, st, a, b, WGS, it, bad);
        } else {
          hipExtLaunchKernelGGL(spin_kernel, dim3(WGS), dim3(256), 0, st, nullptr, nullptr, 0, ticks, a, it);
          hipExtLaunchKernelGGL(spin_kernel, dim3(WGS), dim3(256), 0, st, nullptr, nullptr, mode ? hipExtAnyOrderLaunch : 0, ticks, b, it);
          hipExtLaunchKernelGGL(check_kernel, dim3(1), dim3(64), 0, st, nullptr, nullptr, mode == 3 ? hipExtAnyOrderLaunch : 0, a, b, WGS, it, bad);
        }
      }
      CHECK(hipEventRecord(e1, st)); CHECK(hipEventSynchronize(e1));
      CHECK(hipStreamSynchronize(st2));
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      CHECK(hipMemcpy(&hbad, bad, 4, hipMemcpyDeviceToHost));
    }
    printf("%-52s %7.2f us per A,B,C chain (A = B = %d us)  order violations seen by C: %d\n", names[mode], ms * 1e3 / iters, us, hbad);
  }
  return 0;
}
