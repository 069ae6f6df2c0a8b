// Micro-benchmark: what the instructions that move data between lanes or through LDS cost on gfx950 with the fused
// kernel's occupancy (three workgroups of four waves per CU): ds_bpermute_b32, ds_swizzle_b32, the stores and loads of
// the transform's exchange and of the spectrum rows, and the two data-parallel moves that replaced the bpermute.
// Printed: cycles per wave-instruction per CU at 2.4 GHz (all twelve waves of a CU issue the same stream).
#include <hip/hip_runtime.h>
#include <cstdio>

template <int KIND>
__global__ void __launch_bounds__(256) k(float* out, int iters, float seed) {
  extern __shared__ float lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  typedef float f2 __attribute__((ext_vector_type(2)));
  typedef float f4 __attribute__((ext_vector_type(4)));
  float v[16];
  f2 v2[8];
  f4 v4[4];
#pragma unroll
  for (int i = 0; i < 16; ++i) v[i] = seed + i + threadIdx.x;
#pragma unroll
  for (int i = 0; i < 8; ++i) v2[i] = f2{seed + i, seed - i};
#pragma unroll
  for (int i = 0; i < 4; ++i) v4[i] = f4{seed + i, seed - i, seed, seed * 2};
  const int paddr = ((lane & 48) | ((16 - (lane & 15)) & 15)) << 2;
  const unsigned base = (unsigned)(wave * 8192 + lane * 8);          // 8-byte slots, conflict-free
  const unsigned base16 = (unsigned)(wave * 8192 + lane * 16);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (KIND == 0) asm volatile("ds_bpermute_b32 %0, %1, %0" : "+v"(v[i]) : "v"(paddr));
      if (KIND == 1) asm volatile("ds_swizzle_b32 %0, %0 offset:swizzle(BITMASK_PERM, \"0000p\")" : "+v"(v[i]));
      if (KIND == 2) asm volatile("ds_write_b64 %0, %1 offset:%2" :: "v"(base), "v"(v2[i & 7]), "i"(i * 512) : "memory");
      if (KIND == 3) asm volatile("ds_write_b32 %0, %1 offset:%2" :: "v"(base), "v"(v[i]), "i"(i * 256) : "memory");
      if (KIND == 4) asm volatile("ds_write2_b32 %0, %1, %2 offset0:%3 offset1:%4" :: "v"(base), "v"(v[i]), "v"(v[(i + 1) & 15]), "i"((i * 2) & 255), "i"((i * 2 + 64) & 255) : "memory");
      if (KIND == 5) asm volatile("ds_write_b128 %0, %1 offset:%2" :: "v"(base16), "v"(v4[i & 3]), "i"(i * 1024 & 0xffff) : "memory");
      if (KIND == 6) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v4[i & 3]) : "v"(base16), "i"((i * 1024) & 0x3fff) : "memory");
      if (KIND == 7) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v2[i & 7]) : "v"(base), "i"(i * 512) : "memory");
      if (KIND == 8) {
        asm volatile("v_mov_b32_dpp %0, %0 row_ror:15 row_mask:0xf bank_mask:0xf" : "+v"(v[i]));
      }
      if (KIND == 9) {
        asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %0 row_mirror row_mask:0xf bank_mask:0xf" : "+v"(v[i]));
      }
    }
    if (KIND <= 1 || KIND == 6 || KIND == 7) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  float s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += v[i] + v2[i & 7].x + v2[i & 7].y + v4[i & 3].x + v4[i & 3].w;
  if (s == 123.456f) out[threadIdx.x] = s + lds[0];
}

template <int KIND>
void run(const char* name) {
  float* out;
  (void)hipMalloc(&out, 4096);
  const int iters = 4000;
  const size_t smem = 51968;  // three workgroups per CU, as the fused kernel
  (void)hipFuncSetAttribute((const void*)k<KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<KIND><<<256 * 3, 256, smem>>>(out, 10, 1.0f);
  (void)hipDeviceSynchronize();
  float best = 1e9f;
  for (int rep = 0; rep < 5; ++rep) {
    (void)hipEventRecord(e0);
    k<KIND><<<256 * 3, 256, smem>>>(out, iters, 1.0f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    best = ms < best ? ms : best;
  }
  // twelve waves per CU each issue iters * 16 instructions
  printf("%-34s %6.2f cycles per wave-instruction per CU\n", name, best * 1e-3 * 2.4e9 / ((double)iters * 16 * 12));
  (void)hipFree(out);
}

int main() {
  run<8>("(warm-up)");
  run<0>("ds_bpermute_b32");
  run<1>("ds_swizzle_b32");
  run<2>("ds_write_b64");
  run<3>("ds_write_b32");
  run<4>("ds_write2_b32");
  run<5>("ds_write_b128");
  run<6>("ds_read_b128");
  run<7>("ds_read_b64");
  run<8>("v_mov_b32_dpp row_ror:15");
  run<9>("s_nop 1 + v_mov_b32_dpp row_mirror");
  return 0;
}
