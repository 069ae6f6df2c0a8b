// Micro-benchmark of v_mfma_f32_4x4x1_16b_f32 on gfx950 (16 independent 4x4 outer products per
// instruction) beside v_mfma_f32_16x16x4_f32:
//   1. operand layout and exactness: one instruction on a known pattern, compared with fmaf;
//   2. issue cost: cycles per instruction per SIMD at 1 and 3 waves per SIMD;
//   3. does it run beside packed f32 VALU work or take turns with it (as the 16x16x4 form does);
//   4. the bf16 form v_mfma_f32_16x16x32_bf16 beside the same VALU work, for comparison.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));

__global__ void layout_kernel(const float* a, const float* b, float* d) {
  const int lane = threadIdx.x;
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[lane], b[lane], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[64 + lane], b[64 + lane], acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) d[lane * 4 + r] = acc[r];
}

// KIND: 0 = 4x4x1 f32, 1 = 16x16x4 f32, 2 = 16x16x32 bf16
template <int M, int V, int KIND>
__global__ void __launch_bounds__(256) k(float* out, int iters, float seed) {
  f4 acc[4];
  f2 b[8];
#pragma unroll
  for (int i = 0; i < 4; ++i) acc[i] = f4{seed, seed, seed, seed};
#pragma unroll
  for (int i = 0; i < 8; ++i) b[i] = f2{seed + i, seed - i};
  const float c = seed * 0.5f;
  const f2 c2 = {c, c};
  f4 cb = {c, c, c, c};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int i = 0; i < M / 4; ++i) {
        if (KIND == 0) asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0" : "+v"(acc[i & 3]) : "v"(c), "v"(c));
        if (KIND == 1) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[i & 3]) : "v"(c), "v"(c));
        if (KIND == 2) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i & 3]) : "v"(cb), "v"(cb));
      }
#pragma unroll
      for (int i = 0; i < V / 4; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(b[i & 7]) : "v"(c2));
    }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += b[i].x + b[i].y;
  if (s == 123.456f) out[threadIdx.x] = s;
}

template <int M, int V, int KIND>
float run(const char* name, int w) {
  float* out;
  hipMalloc(&out, 4096);
  const int iters = 2000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) k<M, V, KIND><<<256 * w, 256>>>(out, iters, 1.0f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<M, V, KIND><<<256 * w, 256>>>(out, iters, 1.0f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  printf("%-40s waves/SIMD=%d  %.3f ms  (%.1f cycles per round per wave at 2.4 GHz)\n", name, w, ms, ms * 1e-3 * 2.4e9 / iters / w);
  hipFree(out);
  return ms;
}

int main() {
  {
    float ha[128], hb[128], hd[256], *a, *b, *d;
    for (int i = 0; i < 128; ++i) { ha[i] = 1.0f + 0.37f * i + 1e-3f * (i % 7); hb[i] = 0.5f - 0.11f * i + 1e-4f * (i % 5); }
    hipMalloc(&a, sizeof(ha)); hipMalloc(&b, sizeof(hb)); hipMalloc(&d, sizeof(hd));
    hipMemcpy(a, ha, sizeof(ha), hipMemcpyHostToDevice);
    hipMemcpy(b, hb, sizeof(hb), hipMemcpyHostToDevice);
    layout_kernel<<<1, 64>>>(a, b, d);
    hipMemcpy(hd, d, sizeof(hd), hipMemcpyDeviceToHost);
    // hypothesis: lane l = 4 blk + x supplies A[blk][row x] and B[blk][col x]; register r of lane l holds D[blk][row r][col x]
    int bad = 0;
    for (int l = 0; l < 64; ++l)
      for (int r = 0; r < 4; ++r) {
        const int blk = l / 4, x = l % 4;
        const float want = fmaf(ha[64 + 4 * blk + r], hb[64 + 4 * blk + x], ha[4 * blk + r] * hb[4 * blk + x]);
        if (hd[l * 4 + r] != want) {
          if (bad < 5) printf("layout mismatch lane %d reg %d: got %.9g want %.9g\n", l, r, hd[l * 4 + r], want);
          ++bad;
        }
      }
    printf("4x4x1_16b layout (lane = 4 block + index; D[row = register][col = lane %% 4]) and fmaf-chain exactness: %s (%d mismatches)\n",
           bad ? "DIFFERENT" : "confirmed", bad);
  }
  for (int w : {1, 3}) {
    run<16, 0, 0>("16 mfma 4x4x1_16b f32", w);
    run<8, 0, 1>("8 mfma 16x16x4 f32", w);
    run<16, 0, 2>("16 mfma 16x16x32 bf16", w);
    run<0, 48, 0>("48 pk_fma", w);
    run<16, 48, 0>("16 mfma 4x4x1 + 48 pk_fma", w);
    run<8, 48, 1>("8 mfma 16x16x4 + 48 pk_fma", w);
    run<16, 48, 2>("16 mfma 16x16x32 bf16 + 48 pk_fma", w);
  }
  return 0;
}
