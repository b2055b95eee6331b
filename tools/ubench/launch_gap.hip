// Micro-benchmark: is the boundary between two dependent launches cheaper inside a hipGraph than on a stream?
// P launches of one kernel, phase p reads what phase p - 1 wrote; (a) enqueued one by one on a stream, (b) the same P launches
// captured once into a graph and replayed.  hipcc --offload-arch=gfx950 -O3 launch_gap.hip -o launch_gap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(1024) void phase_kernel(const float* in, float* out, int n_per_wg) {
  const int src = (blockIdx.x + 37) % gridDim.x;
  for (int i = threadIdx.x; i < n_per_wg; i += blockDim.x) out[(size_t)blockIdx.x * n_per_wg + i] = in[(size_t)src * n_per_wg + i] + 1.f;
}

int main(int argc, char** argv) {
  const int phases = argc > 1 ? atoi(argv[1]) : 200;
  const int WGS = 256;
  float *a, *b;
  const int max_per_wg = 64 * 1024 / 4;
  CHECK(hipMalloc(&a, (size_t)WGS * max_per_wg * 4)); CHECK(hipMalloc(&b, (size_t)WGS * max_per_wg * 4));
  hipStream_t st; CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int kb : {0, 4, 16}) {
    int n_per_wg = kb ? kb * 1024 / 4 : 1;
    std::vector<float> host((size_t)WGS * n_per_wg);
    auto enqueue = [&]() {
      for (int p = 0; p < phases; ++p)
        hipLaunchKernelGGL(phase_kernel, dim3(WGS), dim3(1024), 0, st, (p & 1) ? b : a, (p & 1) ? a : b, n_per_wg);
    };
    hipGraph_t graph; hipGraphExec_t exec;
    CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    enqueue();
    CHECK(hipStreamEndCapture(st, &graph));
    CHECK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    float ms[2] = {0, 0};
    bool ok[2] = {true, true};
    for (int mode = 0; mode < 2; ++mode) {
      for (int rep = 0; rep < 4; ++rep) {
        CHECK(hipMemsetAsync(a, 0, (size_t)WGS * n_per_wg * 4, st)); CHECK(hipMemsetAsync(b, 0, (size_t)WGS * n_per_wg * 4, st));
        CHECK(hipStreamSynchronize(st));
        CHECK(hipEventRecord(e0, st));
        if (mode == 0) enqueue(); else CHECK(hipGraphLaunch(exec, st));
        CHECK(hipEventRecord(e1, st)); CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&ms[mode], e0, e1));
      }
      CHECK(hipMemcpy(host.data(), (phases & 1) ? b : a, host.size() * 4, hipMemcpyDeviceToHost));
      for (float v : host) if (v != (float)phases) { ok[mode] = false; break; }
    }
    printf("%3d KiB per workgroup and phase: stream launches %6.2f us/phase %s | graph replay %6.2f us/phase %s\n", kb,
           ms[0] * 1e3 / phases, ok[0] ? "ok" : "WRONG", ms[1] * 1e3 / phases, ok[1] ? "ok" : "WRONG");
    CHECK(hipGraphExecDestroy(exec)); CHECK(hipGraphDestroy(graph));
  }
  return 0;
}
