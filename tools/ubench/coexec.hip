// Micro-benchmark: do fp32 MFMA and (packed) fp32 VALU instructions of gfx950 execute side by side?
// Every wave runs `iters` rounds of M independent v_mfma_f32_16x16x4_f32 and/or V VALU
// instructions of one kind; the grid puts W waves on every SIMD.  If the two pipes run side by
// side, time(both) ~ max(time(mfma), time(valu)); if they share one, time(both) ~ the sum.
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <int M, int V, int KIND>
__global__ void __launch_bounds__(256) k(float* out, int iters, float seed) {
  f4 acc[4];
  f2 b[8];
  float a[8];
#pragma unroll
  for (int i = 0; i < 4; ++i) acc[i] = f4{seed, seed, seed, seed};
#pragma unroll
  for (int i = 0; i < 8; ++i) { b[i] = f2{seed + i, seed - i}; a[i] = seed + i; }
  const float c = seed * 0.5f;
  const f2 c2 = {c, c};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int i = 0; i < M / 4; ++i) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[i & 3]) : "v"(c), "v"(c));
#pragma unroll
      for (int i = 0; i < V / 4; ++i) {
        if (KIND == 0) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(b[i & 7]) : "v"(c2));
        if (KIND == 1) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i & 7]) : "v"(c));
        if (KIND == 2) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i & 7]) : "v"(c));
        if (KIND == 3) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(b[i & 7]) : "v"(c2));
      }
    }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += b[i].x + b[i].y + a[i];
  if (s == 123.456f) out[threadIdx.x] = s;
}

template <int M, int V, int KIND>
float run(const char* name, int w) {
  float* out;
  hipMalloc(&out, 4096);
  const int iters = 2000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<M, V, KIND><<<256 * w, 256>>>(out, 10, 1.0f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<M, V, KIND><<<256 * w, 256>>>(out, iters, 1.0f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  printf("%-34s waves/SIMD=%d  %.3f ms  (%.1f cycles per round per SIMD at 2.4 GHz)\n", name, w, ms, ms * 1e-3 * 2.4e9 / iters / w);
  hipFree(out);
  return ms;
}

int main() {
  for (int w : {1, 3}) {
    run<8, 0, 0>("8 mfma", w);
    run<0, 48, 0>("48 pk_fma", w);
    run<8, 48, 0>("8 mfma + 48 pk_fma", w);
    run<0, 48, 1>("48 v_fma", w);
    run<8, 48, 1>("8 mfma + 48 v_fma", w);
    run<0, 48, 2>("48 v_add", w);
    run<8, 48, 2>("8 mfma + 48 v_add", w);
    run<0, 48, 3>("48 pk_add", w);
    run<8, 48, 3>("8 mfma + 48 pk_add", w);
  }
  return 0;
}
