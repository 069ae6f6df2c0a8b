// Micro-benchmark: what store rate does HBM reach for the write patterns of the stacked-spectrum
// output [frames, 17 * 257] float32 (52 GB for 10 000 x 299 frames)?
//   seq      every workgroup fills one utterance's 299 * 17 476 bytes front to back (16 bytes per lane)
//   row      a wave writes one whole output row (17 476 contiguous bytes), rows of a batch of 16
//            over the 4 waves
//   runs     a wave takes an output row and writes, back to back, the blocks of it that the 16 frames
//            of the current batch supply (1 to 16 adjacent blocks: each row is written in two runs)
//   row by blocks  the same row written as the 63 or 64 whole chunks of each 257-float block, then the
//            16 chunks that straddle two blocks
//   scatter  a wave writes the 1028 bytes of a source frame to its 17 places (row t - 8 + j,
//            block 16 - j), 16 448 bytes apart: the order the fused kernel uses
// All patterns write the same bytes (aligned down to 16); data comes from registers.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int B = 257, N = 17, F = 299;
constexpr long W = (long)N * B;  // floats per output row

// MODE 4 = scatter with the real kernel's alignment handling: whole 16-byte chunks where the block's
// phase allows, single floats at its head and tail.  Dynamic LDS only limits the workgroups per CU.
template <int MODE, bool NT>
__global__ void __launch_bounds__(256) k(float* out) {
  extern __shared__ float dyn[];
  if (out == nullptr) dyn[threadIdx.x] = 0.f;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* const o = out + (long)blockIdx.x * F * W;
  const f4 v = {1.f + lane, 2.f, 3.f, 4.f + wave};
  auto st = [&](float* p) {
    f4* q = (f4*)((uintptr_t)p & ~(uintptr_t)15);
    if (NT) __builtin_nontemporal_store(v, q); else *q = v;
  };
  if (MODE == 0) {
    const long n16 = (long)F * W / 4;
    for (long i = threadIdx.x; i < n16; i += 256) st(o + 4 * i);
  } else if (MODE == 1) {
    for (int tb = 0; tb < F; tb += 16)
      for (int i = wave; i < 16 && tb + i < F; i += 4) {
        float* const row = o + (long)(tb + i) * W;
        for (int c = lane; c < (int)((W + 3) / 4); c += 64) st(row + 4 * c);
      }
  } else if (MODE == 3) {
    for (int tb = 0; tb < F; tb += 16)
      for (int t = tb - 8 + wave; t < tb + 24; t += 4) {
        if (t < 0 || t >= F) continue;
        const int s_lo = max(tb, t - 8), s_hi = min(min(tb + 15, t + 8), F - 1);
        for (int s = s_lo; s <= s_hi; ++s) {
          float* const d = o + (long)t * W + (long)(s - t + 8) * B;
          st(d + 4 * lane);
          if (lane == 0) st(d + 256);
        }
      }
  } else if (MODE == 5) {
    // a row block by block: the whole chunks inside each 257-float block, then the chunks across the
    // block boundaries in one more instruction (16 lanes, 1028 bytes apart)
    for (int tb = 0; tb < F; tb += 16)
      for (int i = wave; i < 16 && tb + i < F; i += 4) {
        const int t = tb + i;
        const int a = (int)(((long)blockIdx.x * F + t) & 3);
        float* const row = o + (long)t * W;
        for (int f = 0; f < N; ++f) {
          const int c0 = (4 - ((a + f) & 3)) & 3, nbody = (B - c0) >> 2;
          if (lane < nbody) *(f4*)(row + f * B + c0 + 4 * lane) = v;
        }
        if (lane < N) {
          const int c0 = (4 - ((a + lane) & 3)) & 3;
          if (c0 > 0 && (t > 0 || lane > 0)) *(f4*)(row + lane * B + c0 - 4) = v;
        }
      }
  } else if (MODE == 4) {
    for (int tb = 0; tb < F; tb += 16)
      for (int i = wave; i < 16 && tb + i < F; i += 4) {
        const int s = tb + i;
        const int a = (int)(((long)blockIdx.x * F + s + 8) & 3);
        const int c0 = (4 - a) & 3, nbody = (B - c0) >> 2, ntail = B - c0 - 4 * nbody;
        const bool body = lane < nbody, edge = lane < c0 + ntail;
        const int ce = lane < c0 ? lane : 4 * nbody + lane;
        for (int j = 0; j < N; ++j) {
          const int t = s - 8 + j;
          if (t < 0 || t >= F) continue;
          float* const d = o + (long)t * W + (long)(N - 1 - j) * B;
          if (body) *(f4*)(d + c0 + 4 * lane) = v;
          if (edge) d[ce] = v.x;
        }
      }
  } else {
    for (int tb = 0; tb < F; tb += 16)
      for (int i = wave; i < 16 && tb + i < F; i += 4) {
        const int s = tb + i;
        for (int j = 0; j < N; ++j) {
          const int t = s - 8 + j;
          if (t < 0 || t >= F) continue;
          float* const d = o + (long)t * W + (long)(N - 1 - j) * B;
          st(d + 4 * lane);
          if (lane == 0) st(d + 256);
        }
      }
  }
}

template <int MODE, bool NT>
void run(const char* name, float* out, int utts, int lds = 0) {
  hipFuncSetAttribute((const void*)k<MODE, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE, NT><<<utts, 256, lds>>>(out);
  hipDeviceSynchronize();
  float best = 1e9f;
  for (int r = 0; r < 3; ++r) {
    hipEventRecord(e0);
    k<MODE, NT><<<utts, 256, lds>>>(out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    best = ms < best ? ms : best;
  }
  const double bytes = (double)utts * F * W * 4;
  printf("%-28s %8.3f ms  %7.1f GB/s\n", name, best, bytes / best / 1e6);
}

int main(int argc, char** argv) {
  const int utts = argc > 1 ? atoi(argv[1]) : 10000;
  float* out;
  if (hipMalloc(&out, (size_t)utts * F * W * 4 + 64) != hipSuccess) { printf("alloc failed\n"); return 1; }
  run<0, false>("seq", out, utts);
  run<0, true>("seq nontemporal", out, utts);
  run<1, false>("row", out, utts);
  run<1, true>("row nontemporal", out, utts);
  run<3, false>("runs", out, utts);
  run<2, false>("scatter", out, utts);
  run<2, true>("scatter nontemporal", out, utts);
  run<2, false>("scatter, 3 workgroups/CU", out, utts, 53536);
  run<4, false>("scatter + edges", out, utts);
  run<4, false>("scatter + edges, 3 wg/CU", out, utts, 53536);
  run<1, false>("row, 3 workgroups/CU", out, utts, 53536);
  run<1, false>("row, 2 workgroups/CU", out, utts, 65536);
  run<5, false>("row by blocks, 2 wg/CU", out, utts, 65536);
  run<3, false>("runs, 3 workgroups/CU", out, utts, 53536);
  hipFree(out);
  return 0;
}
