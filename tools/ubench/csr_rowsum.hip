// SURVEY §8(f) row 3, measured: the sparse-aware real path E_real[b,:] = sum_{n in row b} We[n,:] (X is binary)
// as a CSR row-sum gather, to be compared with the dense MFMA GEMM that computes the same rows
// (tools/gemm_bench.py, NN 128 x e x N).  One workgroup per (row, 256-column slab); float4 per lane;
// CSR order, so the sum is deterministic.  Build: hipcc -O3 --offload-arch=gfx950 csr_rowsum.hip -o csr_rowsum
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ __launch_bounds__(64) void csr_rowsum(const long long* __restrict__ indptr, const int* __restrict__ indices,
                                                 const float* __restrict__ W, int ldw, float* __restrict__ out, int ldo) {
  const int row = blockIdx.y, c4 = blockIdx.x * 64 + threadIdx.x;     // float4 column
  if (c4 * 4 >= ldw) return;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  const long long s = indptr[row], e = indptr[row + 1];
  long long j = s;
  for (; j + 4 <= e; j += 4) {                                        // four independent gathers in flight
    const int n0 = indices[j], n1 = indices[j + 1], n2 = indices[j + 2], n3 = indices[j + 3];
    const float4 a = *reinterpret_cast<const float4*>(W + (size_t)n0 * ldw + 4 * c4);
    const float4 b = *reinterpret_cast<const float4*>(W + (size_t)n1 * ldw + 4 * c4);
    const float4 c = *reinterpret_cast<const float4*>(W + (size_t)n2 * ldw + 4 * c4);
    const float4 d = *reinterpret_cast<const float4*>(W + (size_t)n3 * ldw + 4 * c4);
    acc.x += a.x; acc.y += a.y; acc.z += a.z; acc.w += a.w;
    acc.x += b.x; acc.y += b.y; acc.z += b.z; acc.w += b.w;
    acc.x += c.x; acc.y += c.y; acc.z += c.z; acc.w += c.w;
    acc.x += d.x; acc.y += d.y; acc.z += d.z; acc.w += d.w;
  }
  for (; j < e; ++j) {
    const float4 a = *reinterpret_cast<const float4*>(W + (size_t)indices[j] * ldw + 4 * c4);
    acc.x += a.x; acc.y += a.y; acc.z += a.z; acc.w += a.w;
  }
  *reinterpret_cast<float4*>(out + (size_t)row * ldo + 4 * c4) = acc;
}

static void run(const char* name, int B, int N, int e, double density) {
  const int ld = (e + 63) / 64 * 64;
  std::vector<long long> indptr(B + 1, 0);
  std::vector<int> indices;
  srand(7);
  for (int b = 0; b < B; ++b) {
    // log-normal-ish activity: between 0.2x and 5x the mean
    const double f = 0.2 + 4.8 * (rand() / (double)RAND_MAX) * (rand() / (double)RAND_MAX);
    const int nnz = (int)(density * N * f / 1.4) + 1;
    for (int k = 0; k < nnz; ++k) indices.push_back((int)((long long)rand() * 7919 % N));
    indptr[b + 1] = (long long)indices.size();
  }
  long long* d_ip; int* d_ix; float *W, *out;
  hipMalloc(&d_ip, (B + 1) * 8); hipMalloc(&d_ix, indices.size() * 4);
  hipMalloc(&W, (size_t)N * ld * 4); hipMalloc(&out, (size_t)B * ld * 4);
  hipMemcpy(d_ip, indptr.data(), (B + 1) * 8, hipMemcpyHostToDevice);
  hipMemcpy(d_ix, indices.data(), indices.size() * 4, hipMemcpyHostToDevice);
  hipMemset(W, 0, (size_t)N * ld * 4);
  hipEvent_t a, b2; hipEventCreate(&a); hipEventCreate(&b2);
  dim3 grid((ld / 4 + 63) / 64, B);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(csr_rowsum, grid, dim3(64), 0, 0, d_ip, d_ix, W, ld, out, ld);
  hipEventRecord(a);
  const int iters = 20;
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(csr_rowsum, grid, dim3(64), 0, 0, d_ip, d_ix, W, ld, out, ld);
  hipEventRecord(b2); hipEventSynchronize(b2);
  float ms; hipEventElapsedTime(&ms, a, b2);
  const double us = ms / iters * 1e3, bytes = (double)indices.size() * ld * 4;
  printf("%-28s B=%d N=%d e=%d nnz/row=%.0f: %7.1f us  gathered %.1f MB at %.2f TB/s   (dense half = %.2f GFLOP)\n", name, B, N, e,
         indices.size() / (double)B, us, bytes / 1e6, bytes / us / 1e6, 2.0 * B * N * e / 1e9);
  hipFree(d_ip); hipFree(d_ix); hipFree(W); hipFree(out);
}

int main() {
  run("C2 ML-1M (3.5 % dense)", 128, 3706, 992, 0.035);
  run("C1 LastFM (0.25 %)", 32, 17632, 32, 0.0025);
  run("C4 shard (1 %), e=1024", 128, 50000, 1024, 0.01);
  run("C4 shard (1 %), e=32", 128, 50000, 32, 0.01);
  return 0;
}
