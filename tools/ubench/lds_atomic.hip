// Micro-benchmark: cost of float adds into LDS on gfx950 -- ds_add_f32 (no result) with all 64 lanes on different
// words, with four lanes per word, with 13 or 16 active lanes, against a plain read / add / write of the same words.
// 256 blocks x 12 waves (three per SIMD), each wave adds into its own 256 bytes; prints nanoseconds and cycles
// (at 2.4 GHz) per wave-instruction per CU.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int KIND>
__global__ void __launch_bounds__(768) k(float* out, int iters) {
  __shared__ float buf[12 * 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* const mine = buf + wave * 64;
  mine[lane] = 0.f;
  __syncthreads();
  float x = (float)lane;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (KIND == 0) __hip_atomic_fetch_add(mine + lane, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (KIND == 1) __hip_atomic_fetch_add(mine + (lane & 15), x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (KIND == 2) { if (lane < 16) __hip_atomic_fetch_add(mine + lane, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
      if (KIND == 3) { if ((lane & 15) < 13) __hip_atomic_fetch_add(mine + (lane & 15), x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
      if (KIND == 4) { float v = mine[lane]; asm volatile("" : "+v"(v)); mine[lane] = v + x; }
      if (KIND == 5) __hip_atomic_fetch_add(mine + (lane >> 2), x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (KIND == 6) { if (lane < 13) { float v = mine[lane]; asm volatile("" : "+v"(v)); mine[lane] = v + x; } }
      asm volatile("" : "+v"(x));
    }
  }
  __syncthreads();
  if (mine[lane] == 123.456f) out[threadIdx.x] = x;
}

template <int KIND>
void run(const char* name) {
  float* out;
  (void)hipMalloc(&out, 4096);
  const int iters = 2000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<KIND><<<256, 768>>>(out, 10);
  (void)hipDeviceSynchronize();
  float best = 1e9f;
  for (int rep = 0; rep < 5; ++rep) {
    (void)hipEventRecord(e0);
    k<KIND><<<256, 768>>>(out, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    best = ms < best ? ms : best;
  }
  const double per = best * 1e6 / (iters * 8.0 * 12);  // ns per wave-instruction per CU
  printf("%-52s %.3f ms  %.2f ns = %.1f cycles per wave-instruction per CU\n", name, best, per, per * 2.4);
  (void)hipFree(out);
}

int main() {
  for (int i = 0; i < 10; ++i) run<4>("(warm-up)");
  run<0>("ds_add_f32, 64 lanes, 64 words");
  run<5>("ds_add_f32, 64 lanes, 16 words (4 neighbours each)");
  run<1>("ds_add_f32, 64 lanes, 16 words (lanes 16 apart)");
  run<3>("ds_add_f32, 52 lanes, 13 words (lanes 16 apart)");
  run<2>("ds_add_f32, 16 lanes, 16 words");
  run<4>("read + add + write, 64 lanes");
  run<6>("read + add + write, 13 lanes");
  return 0;
}
