// Micro-benchmark: issue cost of plain, packed and 64-bit vector instructions on gfx950 as a function of the
// number of waves a SIMD hosts (1 .. 8): W workgroups of four waves per CU (their dynamic LDS keeps more out),
// every wave runs blocks of 16 independent instructions; printed: cycles per wave-instruction per SIMD at 2.4 GHz.
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f2 __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ void k(float* out, int iters, float seed) {
  extern __shared__ float lds[];
  float a[16];
  f2 b[16];
  double d[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) { a[i] = seed + i + threadIdx.x; b[i] = f2{seed + i, seed - i}; d[i] = seed + i; }
  const float c = seed * 0.5f, e = seed * 0.25f;
  const f2 c2 = {c, e};
  const double dc = seed * 0.5;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (KIND == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
      if (KIND == 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(e));
      if (KIND == 2) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(b[i]) : "v"(c2));
      if (KIND == 3) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(b[i]) : "v"(c2));
      if (KIND == 4) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d[i]) : "v"(dc));
      if (KIND == 5) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(dc));
      if (KIND == 6) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(b[i]) : "v"(c2));
      if (KIND == 7) asm volatile("v_mov_b64 %0, %1" : "=v"(d[i]) : "v"(dc));
      if (KIND == 8) asm volatile("v_pk_fma_f16 %0, %0, %1, %1" : "+v"(a[i]) : "v"(c));
      if (KIND == 9) asm volatile("v_pk_mov_b32 %0, %1, %1" : "=v"(b[i]) : "v"(c2));
      // source operand count of a packed fma: three different register pairs; two pairs and a scalar pair; the
      // modifiers of a complex product's second half; a chain in which every instruction needs the one before it
      if (KIND == 12) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(b[i]) : "v"(b[(i + 5) & 15]), "v"(c2));
      if (KIND == 13) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(b[i]) : "s"(c2), "v"(b[(i + 5) & 15]));
      if (KIND == 14) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "+v"(b[i]) : "v"(b[(i + 5) & 15]), "v"(c2));
      if (KIND == 15) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(b[(i + 1) & 15]) : "v"(b[i]), "v"(c2));
      if (KIND == 16) asm volatile("v_pk_add_f32 %0, %1, %2" : "+v"(b[i]) : "v"(b[(i + 5) & 15]), "v"(b[(i + 9) & 15]));
      if (KIND == 17) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(b[i]) : "v"(b[(i + 5) & 15]));     // result over its first source
      if (KIND == 18) asm volatile("v_pk_add_f32 %0, %1, %2" : "+v"(b[i]) : "v"(b[(i + 5) & 15]), "v"(c2));  // one source shared by all
      if (KIND == 19) asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "+v"(b[i]) : "v"(b[(i + 5) & 15]), "s"(c2), "v"(b[(i + 9) & 15]));
      if (KIND == 20) asm volatile("v_pk_mul_f32 %0, %1, %2" : "+v"(b[i]) : "v"(b[(i + 5) & 15]), "v"(b[(i + 9) & 15]));
      if (KIND == 10) {  // the transform's mix: packed and plain alternate
        if (i & 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(b[i]) : "v"(c2));
        else asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
      }
      if (KIND == 11) {  // three packed, one plain
        if ((i & 3) != 3) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(b[i]) : "v"(c2));
        else asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
      }
    }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += a[i] + b[i].x + b[i].y + (float)d[i];
  if (s == 123.456f) out[threadIdx.x] = s + lds[0];
}

template <int KIND>
void run(const char* name) {
  float* out;
  (void)hipMalloc(&out, 4096);
  const int iters = 2000;
  (void)hipFuncSetAttribute((const void*)k<KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  printf("%-14s", name);
  for (int w : {1, 2, 3, 4, 5, 6, 8}) {
    // w workgroups of four waves per CU: each takes a w-th of the CU's 160 KB of LDS, so no more fit
    const size_t smem = (size_t)(160 * 1024 / w) & ~(size_t)1279;
    k<KIND><<<256 * w, 256, smem>>>(out, 10, 1.0f);
    (void)hipDeviceSynchronize();
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
      (void)hipEventRecord(e0);
      k<KIND><<<256 * w, 256, smem>>>(out, iters, 1.0f);
      (void)hipEventRecord(e1);
      (void)hipEventSynchronize(e1);
      float ms;
      (void)hipEventElapsedTime(&ms, e0, e1);
      best = ms < best ? ms : best;
    }
    printf("  w=%d %5.2f", w, best * 1e-3 * 2.4e9 / ((double)iters * 16 * w));
  }
  printf("\n");
  (void)hipFree(out);
}

int main() {
  for (int i = 0; i < 3; ++i) run<0>("(warm-up)");
  run<0>("v_add_f32");
  run<1>("v_fma_f32");
  run<2>("v_pk_fma_f32");
  run<3>("v_pk_add_f32");
  run<6>("v_pk_mul_f32");
  run<4>("v_fma_f64");
  run<5>("v_add_f64");
  run<7>("v_mov_b64");
  run<9>("v_pk_mov_b32");
  run<8>("v_pk_fma_f16");
  run<12>("pk_fma 3 vgpr");
  run<13>("pk_fma 2v+s");
  run<14>("pk_fma 3v mod");
  run<15>("pk_fma chain");
  run<16>("pk_add 2 src");
  run<17>("pk_add d=s0");
  run<18>("pk_add shared");
  run<19>("pk_fma v,s,v");
  run<20>("pk_mul 2 src");
  run<10>("pk/plain 1:1");
  run<11>("pk/plain 3:1");
  return 0;
}
