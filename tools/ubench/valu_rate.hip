// Issue cost of the vector instructions of the bf16 split, per wave and per SIMD (gfx950).
// One workgroup of 64 / 256 / 1024 threads (one wave; one wave on each SIMD; four waves on each SIMD) runs REPS x 32 instances of one
// instruction -- either 32 independent ones (throughput) or one dependent chain (latency) -- between two s_memtime stamps; the table is
// s_memtime ticks (a constant-rate counter, not the shader clock: the ratio to v_sub_f32 is what to read) per instruction
// of ONE wave, i.e. with four waves per SIMD a port that serves them in turn shows 4 x its issue cost.
// build: hipcc --offload-arch=gfx950 -O2 tools/ubench/valu_rate.hip -o tools/ubench/valu_rate ; run: tools/ubench/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int REPS = 256;

enum Op { SUB = 0, PK_ADD, CVT_BF16, LSHL, AND, PERM, FMA, PK_FMA, CVT_F16, ADD_F32, MUL_F32, FMAC, ADD_U32, OR_B32, LSHR, LSHL16, MUL_U24, MOV, MOV_SDWA,
          CVT_F32_BF16, CVT_F32_BF16_SDWA, AND_REG, BFI, AND_OR, MAX_F32, SUB_E64, LSHL_ADD, LSHL_OR, ALIGNBIT, MAD_U24, ASHR, XOR_B32, MIN_F32, BFE, LSHL_REG, MUL_LO, NOPS };
static const char* kName[] = {"v_sub_f32", "v_pk_add_f32", "v_cvt_pk_bf16_f32", "v_lshlrev_b32", "v_and_b32", "v_perm_b32", "v_fma_f32", "v_pk_fma_f32", "v_cvt_pk_f16_f32", "v_add_f32", "v_mul_f32", "v_fmac_f32 (VOP2)", "v_add_u32", "v_or_b32", "v_lshrrev_b32 16", "v_lshlrev_b32 16", "v_mul_u32_u24 65536", "v_mov_b32",
                              "v_mov_b32_sdwa W1<-W0", "v_cvt_f32_bf16", "v_cvt_f32_bf16_sdwa W1", "v_and_b32 (reg mask)", "v_bfi_b32", "v_and_or_b32", "v_max_f32", "v_sub_f32_e64", "v_lshl_add_u32", "v_lshl_or_b32", "v_alignbit_b32", "v_mad_u32_u24", "v_ashrrev_i32 16",
                              "v_xor_b32", "v_min_f32", "v_bfe_u32", "v_lshlrev_b32 (reg shift)", "v_mul_lo_u32"};

template <int OP, bool CHAIN>
__global__ void rate(unsigned long long* out, float seed) {
  float r[32];
  f32x2 q[16];
#pragma unroll
  for (int i = 0; i < 32; ++i) r[i] = seed + i + threadIdx.x;
#pragma unroll
  for (int i = 0; i < 16; ++i) q[i] = f32x2{r[2 * i], r[2 * i + 1]};
  const f32x2 sq = f32x2{seed, seed + 1.f};
  unsigned long long t0, t1;
  __syncthreads();
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int it = 0; it < REPS; ++it) {
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      const int d = CHAIN ? 0 : i, dq = CHAIN ? 0 : (i & 15);
      if constexpr (OP == SUB) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(r[d]) : "v"(seed));
      if constexpr (OP == FMA) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(r[d]) : "v"(seed));
      if constexpr (OP == LSHL) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(r[d]));
      if constexpr (OP == AND) asm volatile("v_and_b32 %0, 0xffff0000, %0" : "+v"(r[d]));
      if constexpr (OP == PERM) asm volatile("v_perm_b32 %0, %0, %1, %1" : "+v"(r[d]) : "v"(seed));
      if constexpr (OP == CVT_BF16) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(r[d]) : "v"(seed));
      if constexpr (OP == CVT_F16) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(r[d]) : "v"(seed));
      if constexpr (OP == ADD_F32) asm volatile("v_add_f32 %0, %0, %1" : "+v"(r[d]) : "v"(seed));
      if constexpr (OP == MUL_F32) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(r[d]) : "v"(seed));
      if constexpr (OP == FMAC) asm volatile("v_fmac_f32 %0, %1, %1" : "+v"(r[d]) : "v"(seed));
      if constexpr (OP == ADD_U32) asm volatile("v_add_u32 %0, %0, %1" : "+v"(r[d]) : "v"(seed));
      if constexpr (OP == OR_B32) asm volatile("v_or_b32 %0, %0, %1" : "+v"(r[d]) : "v"(seed));
      if constexpr (OP == LSHR) asm volatile("v_lshrrev_b32 %0, 16, %0" : "+v"(r[d]));
      if constexpr (OP == LSHL16) asm volatile("v_lshlrev_b32 %0, 16, %0" : "+v"(r[d]));
      if constexpr (OP == MUL_U24) asm volatile("v_mul_u32_u24 %0, 0x10000, %0" : "+v"(r[d]));
      if constexpr (OP == MOV) asm volatile("v_mov_b32 %0, %1" : "+v"(r[d]) : "v"(seed));
      if constexpr (OP == MOV_SDWA) asm volatile("v_mov_b32_sdwa %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PAD src0_sel:WORD_0" : "+v"(r[d]) : "v"(seed));
      if constexpr (OP == CVT_F32_BF16) asm volatile("v_cvt_f32_bf16 %0, %0" : "+v"(r[d]));
      if constexpr (OP == CVT_F32_BF16_SDWA) asm volatile("v_cvt_f32_bf16_sdwa %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "+v"(r[d]));
      if constexpr (OP == AND_REG) asm volatile("v_and_b32 %0, %1, %0" : "+v"(r[d]) : "v"(seed));
      if constexpr (OP == BFI) asm volatile("v_bfi_b32 %0, %1, %0, %1" : "+v"(r[d]) : "v"(seed));
      if constexpr (OP == AND_OR) asm volatile("v_and_or_b32 %0, %0, %1, %1" : "+v"(r[d]) : "v"(seed));
      if constexpr (OP == MAX_F32) asm volatile("v_max_f32 %0, %0, %1" : "+v"(r[d]) : "v"(seed));
      if constexpr (OP == SUB_E64) asm volatile("v_sub_f32_e64 %0, %0, %1" : "+v"(r[d]) : "v"(seed));
      if constexpr (OP == LSHL_ADD) asm volatile("v_lshl_add_u32 %0, %0, 16, 0" : "+v"(r[d]));
      if constexpr (OP == LSHL_OR) asm volatile("v_lshl_or_b32 %0, %0, 16, 0" : "+v"(r[d]));
      if constexpr (OP == ALIGNBIT) asm volatile("v_alignbit_b32 %0, %0, 0, 16" : "+v"(r[d]));
      if constexpr (OP == MAD_U24) asm volatile("v_mad_u32_u24 %0, %0, %1, 0" : "+v"(r[d]) : "v"(seed));
      if constexpr (OP == ASHR) asm volatile("v_ashrrev_i32 %0, 16, %0" : "+v"(r[d]));
      if constexpr (OP == XOR_B32) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(r[d]) : "v"(seed));
      if constexpr (OP == MIN_F32) asm volatile("v_min_f32 %0, %0, %1" : "+v"(r[d]) : "v"(seed));
      if constexpr (OP == BFE) asm volatile("v_bfe_u32 %0, %0, 0, 16" : "+v"(r[d]));
      if constexpr (OP == LSHL_REG) asm volatile("v_lshlrev_b32 %0, %1, %0" : "+v"(r[d]) : "v"(seed));
      if constexpr (OP == MUL_LO) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(r[d]) : "v"(seed));
      if constexpr (OP == PK_ADD) asm volatile("v_pk_add_f32 %0, %0, %1 neg_lo:[0,1] neg_hi:[0,1]" : "+v"(q[dq]) : "v"(sq));
      if constexpr (OP == PK_FMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(q[dq]) : "v"(sq));
    }
  }
  asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  float acc = 0.f;
#pragma unroll
  for (int i = 0; i < 32; ++i) acc += r[i];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc += q[i].x + q[i].y;
  if ((threadIdx.x & 63) == 0) out[threadIdx.x >> 6] = t1 - t0;
  if (acc == 12345.678f) out[63] = 1;   // keeps the registers alive
}

template <int OP, bool CHAIN>
static double run(int threads, unsigned long long* d) {
  std::vector<unsigned long long> h(64);
  double best = 1e30;
  for (int rep = 0; rep < 5; ++rep) {
    hipLaunchKernelGGL((rate<OP, CHAIN>), dim3(1), dim3(threads), 0, 0, d, 1.5f);
    hipMemcpy(h.data(), d, 64 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double worst = 0;
    for (int w = 0; w < threads / 64; ++w) worst = h[w] > worst ? (double)h[w] : worst;
    best = worst < best ? worst : best;
  }
  return best / (REPS * 32.0);
}

template <int OP>
static void row(unsigned long long* d, double base[3]) {
  const int thr[3] = {64, 256, 1024};
  double v[3], c[3];
  for (int i = 0; i < 3; ++i) { v[i] = run<OP, false>(thr[i], d); c[i] = run<OP, true>(thr[i], d); }
  if (OP == SUB) for (int i = 0; i < 3; ++i) base[i] = v[i];
  printf("%-24s independent: %6.3f %6.3f %6.3f  (x v_sub_f32: %4.2f %4.2f %4.2f)   dependent chain: %6.3f %6.3f %6.3f\n", kName[OP], v[0], v[1], v[2],
         v[0] / base[0], v[1] / base[1], v[2] / base[2], c[0], c[1], c[2]);
}

int main() {
  unsigned long long* d;
  hipMalloc(&d, 64 * sizeof(unsigned long long));
  hipMemset(d, 0, 64 * sizeof(unsigned long long));
  double base[3] = {1, 1, 1};
  printf("s_memtime ticks per instruction of one wave; columns: 1 wave, 1 wave per SIMD (4), 4 waves per SIMD (16)\n");
  row<SUB>(d, base); row<FMA>(d, base); row<PK_ADD>(d, base); row<PK_FMA>(d, base); row<CVT_BF16>(d, base); row<CVT_F16>(d, base);
  row<LSHL>(d, base); row<AND>(d, base); row<PERM>(d, base);
  row<ADD_F32>(d, base); row<MUL_F32>(d, base); row<FMAC>(d, base); row<SUB_E64>(d, base); row<MAX_F32>(d, base); row<ADD_U32>(d, base); row<OR_B32>(d, base);
  row<AND_REG>(d, base); row<LSHR>(d, base); row<LSHL16>(d, base); row<MUL_U24>(d, base); row<MOV>(d, base); row<MOV_SDWA>(d, base);
  row<CVT_F32_BF16>(d, base); row<CVT_F32_BF16_SDWA>(d, base); row<BFI>(d, base); row<AND_OR>(d, base);
  row<LSHL_ADD>(d, base); row<LSHL_OR>(d, base); row<ALIGNBIT>(d, base); row<MAD_U24>(d, base); row<ASHR>(d, base); row<XOR_B32>(d, base); row<MIN_F32>(d, base);
  row<BFE>(d, base); row<LSHL_REG>(d, base); row<MUL_LO>(d, base);
  return 0;
}
