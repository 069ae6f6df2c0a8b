// Micro-benchmark: sustained issue rate of a few VALU instruction kinds on gfx950.
// Each wave runs N iterations of 16 independent instructions of one kind; grid fills every
// SIMD with W waves.  Prints cycles per wave-instruction per SIMD (from wall time and 2.4 GHz
// nominal; the ratio between kinds is what matters).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ void __launch_bounds__(256) k(float* out, int iters, float seed) {
  float a[16];
  f2 b[8];
#pragma unroll
  for (int i = 0; i < 16; ++i) a[i] = seed + i + threadIdx.x;
#pragma unroll
  for (int i = 0; i < 8; ++i) b[i] = f2{seed + i, seed - i};
  const float c = seed * 0.5f, d = seed * 0.25f;
  const f2 c2 = {c, d};
  int ia[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) ia[i] = threadIdx.x + i;
  unsigned long long la[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) la[i] = threadIdx.x + i;
  const unsigned long long msk = __builtin_amdgcn_readfirstlane(iters) * 0x123456789ull;
  for (int it = 0; it < iters; ++it) {
    if (KIND == 0) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(d));
    } else if (KIND == 1) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
    } else if (KIND == 2) {
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(b[i]) : "v"(c2));
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(b[i]) : "v"(c2));
    } else if (KIND == 3) {
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(b[i]) : "v"(c2));
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(b[i]) : "v"(c2));
    } else if (KIND == 4) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(ia[i]) : "v"(it));
    } else if (KIND == 5) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(c) : );
    } else if (KIND == 6) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_mov_b32_dpp %0, %1 row_mirror row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(c));
    } else if (KIND == 7) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
    } else if (KIND == 8) {
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(b[i]) : "v"(c2));
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(b[i]) : "v"(c2));
    } else if (KIND == 9) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_log_f32 %0, %0" : "+v"(a[i]));
    } else if (KIND == 10) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "s"(msk));
    } else if (KIND == 11) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(a[i]) : "v"(ia[0]), "v"(c));
    } else if (KIND == 12) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_sub_f32 %0, %1, %0" : "+v"(a[i]) : "v"(c));
    } else if (KIND == 13) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(ia[i]) : "v"(it));
    } else if (KIND == 14) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(ia[i]) : "v"(it));
    } else if (KIND == 15) {
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_lshl_add_u64 %0, %0, 2, %1" : "+v"(la[i]) : "v"(la[8]));
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_lshl_add_u64 %0, %0, 2, %1" : "+v"(la[i]) : "v"(la[8]));
    } else if (KIND == 16) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_cmp_lt_f32 vcc, %0, %1" :: "v"(a[i]), "v"(c) : "vcc");
    } else if (KIND == 17) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
    } else if (KIND == 18) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(a[i]));
    } else if (KIND == 19) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(ia[i]) : "v"(it));
    } else if (KIND == 20) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(d));
    } else if (KIND == 21) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(a[i]) : "v"(c), "v"(d));
    } else if (KIND == 22) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(c));
    } else if (KIND == 23) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "s"(c));
    }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += a[i] + ia[i];
#pragma unroll
  for (int i = 0; i < 8; ++i) s += b[i].x + b[i].y + (float)la[i];
  if (s == 123.456f) out[threadIdx.x] = s;
}

template <int KIND>
void run(const char* name, int waves_per_simd) {
  float* out;
  hipMalloc(&out, 4096);
  const int iters = 4000;
  const int blocks = 256 * waves_per_simd;  // 256-thread blocks = 4 waves = one per SIMD of a CU
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<KIND><<<blocks, 256>>>(out, 10, 1.0f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<KIND><<<blocks, 256>>>(out, iters, 1.0f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double instr_per_simd = (double)iters * 16 * waves_per_simd;  // each SIMD hosts waves_per_simd waves
  const double cyc = ms * 1e-3 * 2.4e9 / instr_per_simd;
  printf("%-14s waves/SIMD=%d  %.3f ms  -> %.2f cycles per wave-instruction per SIMD (at 2.4 GHz)\n", name, waves_per_simd, ms, cyc);
  hipFree(out);
}

int main() {
  for (int w : {3}) {
    run<0>("v_fma_f32", w);
    run<1>("v_add_f32", w);
    run<12>("v_sub_f32", w);
    run<7>("v_mul_f32", w);
    run<23>("v_mul_f32 sgpr", w);
    run<20>("v_fmac_f32", w);
    run<17>("v_max_f32", w);
    run<2>("v_pk_fma_f32", w);
    run<3>("v_pk_add_f32", w);
    run<4>("v_mul_lo_u32", w);
    run<13>("v_add_u32", w);
    run<14>("v_lshl_add_u32", w);
    run<15>("v_lshl_add_u64", w);
    run<19>("v_xor_b32", w);
    run<5>("v_cndmask vcc", w);
    run<21>("v_cndmask nodep", w);
    run<10>("v_cndmask e64", w);
    run<11>("v_bfi_b32", w);
    run<16>("v_cmp_lt_f32", w);
    run<18>("v_cvt_f32_i32", w);
    run<22>("v_mov_b32", w);
    run<6>("v_mov_dpp", w);
    run<9>("v_log_f32", w);
  }
  return 0;
}
