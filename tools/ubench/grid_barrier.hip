// Micro-benchmark: what does a dependent phase boundary cost on this chip?
//   launches   P back-to-back launches of one stream, phase p reads what phase p-1 wrote (another workgroup's lines)
//   flat       ONE cooperative launch of 256 x 1024 threads, P phases separated by a device-wide barrier on one arrival counter
//   xcd        the same with one arrival counter per XCD (workgroup b sits on XCD b % 8) and a second-level counter of eight
// Every phase moves `kb` KiB per workgroup (0 = the boundary alone); the last phase's values prove that every phase saw its
// predecessor's stores.  hipcc --offload-arch=gfx950 -O3 grid_barrier.hip -o grid_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int WGS = 256, THREADS = 1024;

__device__ __forceinline__ void phase_body(const float* __restrict__ in, float* __restrict__ out, int n_per_wg, int wg, int nwg) {
  // workgroup wg reads the slice workgroup (wg + 37) % nwg wrote in the previous phase
  const int src = (wg + 37) % nwg;
  for (int i = threadIdx.x; i < n_per_wg; i += blockDim.x) out[(size_t)wg * n_per_wg + i] = in[(size_t)src * n_per_wg + i] + 1.f;
}

__global__ __launch_bounds__(THREADS) void phase_kernel(const float* in, float* out, int n_per_wg) {
  phase_body(in, out, n_per_wg, blockIdx.x, gridDim.x);
}

// arrival counter that only grows: phase p is over when it reads (p + 1) * arrivals
__device__ __forceinline__ void barrier_flat(unsigned* cnt, unsigned target) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
}

__device__ __forceinline__ void barrier_xcd(unsigned* cnt, unsigned phase, int nwg) {
  // cnt[16 * x] (x < 8): arrivals of XCD x; cnt[16 * 8]: XCDs done (each line of its own)
  __syncthreads();
  if (threadIdx.x == 0) {
    const int x = blockIdx.x & 7;
    const unsigned per = (unsigned)((nwg - x + 7) / 8);
    const unsigned prev = __hip_atomic_fetch_add(cnt + 16 * x, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    if (prev + 1 == (phase + 1) * per) __hip_atomic_fetch_add(cnt + 16 * 8, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    while (__hip_atomic_load(cnt + 16 * 8, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (phase + 1) * 8u) __builtin_amdgcn_s_sleep(1);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
}

template <int MODE>
__global__ __launch_bounds__(THREADS) void persistent_kernel(float* a, float* b, int n_per_wg, int phases, unsigned* cnt) {
  const int nwg = gridDim.x;
  for (int p = 0; p < phases; ++p) {
    const float* in = (p & 1) ? b : a;
    float* out = (p & 1) ? a : b;
    phase_body(in, out, n_per_wg, blockIdx.x, nwg);
    if (MODE == 0) barrier_flat(cnt, (unsigned)(p + 1) * nwg);
    else barrier_xcd(cnt, (unsigned)p, nwg);
  }
}

int main(int argc, char** argv) {
  const int phases = argc > 1 ? atoi(argv[1]) : 200;
  float *a, *b; unsigned* cnt;
  const int max_per_wg = 64 * 1024 / 4;
  CHECK(hipMalloc(&a, (size_t)WGS * max_per_wg * 4)); CHECK(hipMalloc(&b, (size_t)WGS * max_per_wg * 4));
  CHECK(hipMalloc(&cnt, 4096));
  hipStream_t st; CHECK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  int occ = 0;
  CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, persistent_kernel<0>, THREADS, 0));
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
  printf("CUs %d, workgroups of %d threads per CU %d, cooperative launch %d\n", prop.multiProcessorCount, THREADS, occ, prop.cooperativeLaunch);
  if (occ * prop.multiProcessorCount < WGS) { printf("grid does not fit\n"); return 1; }
  for (int kb : {0, 4, 16, 64}) {
    int n_per_wg = kb * 1024 / 4;
    if (n_per_wg == 0) n_per_wg = 1;      // one float per workgroup: the dependence alone
    std::vector<float> host((size_t)WGS * n_per_wg);
    float ms[3] = {0, 0, 0};
    bool ok[3] = {true, true, true};
    for (int mode = 0; mode < 3; ++mode) {
      for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipMemsetAsync(a, 0, (size_t)WGS * n_per_wg * 4, st)); CHECK(hipMemsetAsync(b, 0, (size_t)WGS * n_per_wg * 4, st));
        CHECK(hipMemsetAsync(cnt, 0, 4096, st));
        CHECK(hipEventRecord(e0, st));
        if (mode == 0) {
          for (int p = 0; p < phases; ++p)
            hipLaunchKernelGGL(phase_kernel, dim3(WGS), dim3(THREADS), 0, st, (p & 1) ? b : a, (p & 1) ? a : b, n_per_wg);
        } else {
          int ph = phases;
          void* args[] = {&a, &b, &n_per_wg, &ph, &cnt};
          if (mode == 1) CHECK(hipLaunchCooperativeKernel((const void*)persistent_kernel<0>, dim3(WGS), dim3(THREADS), args, 0, st));
          else CHECK(hipLaunchCooperativeKernel((const void*)persistent_kernel<1>, dim3(WGS), dim3(THREADS), args, 0, st));
        }
        CHECK(hipEventRecord(e1, st)); CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&ms[mode], e0, e1));
      }
      CHECK(hipMemcpy(host.data(), (phases & 1) ? b : a, host.size() * 4, hipMemcpyDeviceToHost));
      for (float v : host) if (v != (float)phases) { ok[mode] = false; break; }
    }
    printf("%3d KiB per workgroup and phase: launches %6.2f us/phase %s | flat barrier %6.2f us/phase %s | per-XCD barrier %6.2f us/phase %s\n", kb,
           ms[0] * 1e3 / phases, ok[0] ? "ok" : "WRONG", ms[1] * 1e3 / phases, ok[1] ? "ok" : "WRONG", ms[2] * 1e3 / phases, ok[2] ? "ok" : "WRONG");
  }
  return 0;
}
