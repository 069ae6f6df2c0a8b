// Host emulation of the fused kernel's spectrum stage for whole utterances:
// reads float32 samples (already pre-emphasised) from stdin-less binary file, writes the
// 257-bin rows (4|X|^2, unscaled) computed with the lane functions of fft_lane.h.
// Usage: spec_sim <in.f32> <n_samples> <out.f32>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "fft_lane.h"
#include "tables.h"

int main(int argc, char** argv) {
  if (argc < 4) return 2;
  const long n = atol(argv[2]);
  std::vector<float> sig(n);
  FILE* f = fopen(argv[1], "rb");
  if (!f || fread(sig.data(), 4, n, f) != (size_t)n) return 3;
  fclose(f);
  const long nfr = mfcc::frames_for_length(n, 400, 160);
  std::vector<float> pad((nfr - 1) * 160 + 400, 0.f);
  for (long i = 0; i < n; ++i) pad[i] = sig[i];
  const auto tw = mfcc::twiddle512();
  std::vector<float> out(nfr * 257);
  for (long t = 0; t < nfr; ++t) {
    float x[512] = {0};
    for (int i = 0; i < 400; ++i) x[i] = pad[t * 160 + i];
    float T[512];
    for (int q = 0; q < 16; ++q) {
      float xr[16], xi[16];
      for (int n1 = 0; n1 < 16; ++n1) { xr[n1] = x[2 * (16 * n1 + q)]; xi[n1] = x[2 * (16 * n1 + q) + 1]; }
      mfcc::fft16(xr, xi);
      for (int k1 = 0; k1 < 16; ++k1) {
        const int m = (2 * q * k1) & 511;
        if (k1) mfcc::cmul(xr[k1], xi[k1], tw[2 * m], tw[2 * m + 1]);
        const int p = mfcc::xpose_index(k1, q);
        T[2 * p] = xr[k1]; T[2 * p + 1] = xi[k1];
      }
    }
    float Zr[16][16], Zi[16][16];
    for (int q = 0; q < 16; ++q) {
      for (int n2 = 0; n2 < 16; ++n2) { const int p = mfcc::xpose_index(q, n2); Zr[q][n2] = T[2 * p]; Zi[q][n2] = T[2 * p + 1]; }
      mfcc::fft16(Zr[q], Zi[q]);
    }
    float* P = &out[t * 257];
    for (int q = 0; q < 16; ++q) {
      const int pl = mfcc::partner_lane(q);
      for (int s = 0; s < 8; ++s) {
        const int k = q + 16 * s;
        float br = Zr[pl][15 - s], bi = Zi[pl][15 - s];
        if (q == 0) { br = Zr[0][(16 - s) & 15]; bi = Zi[0][(16 - s) & 15]; }
        mfcc::untangle_power(Zr[q][s], Zi[q][s], br, bi, tw[2 * k], tw[2 * k + 1], P[k], P[256 - k]);
      }
      if (q == 0) P[128] = 4.f * (Zr[0][8] * Zr[0][8] + Zi[0][8] * Zi[0][8]);
    }
  }
  f = fopen(argv[3], "wb");
  fwrite(out.data(), 4, out.size(), f);
  fclose(f);
  return 0;
}
