// Host emulation of the 16-lane FFT-512 scheme in mfcc_amd/csrc/fft_lane.h:
// runs the lane functions for q = 0..15 against a direct double-precision DFT.
// Usage: fft_sim <frame_size>   (prints max error relative to the spectrum's RMS power)
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "fft_lane.h"
#include "tables.h"

int main(int argc, char** argv) {
  const int frame_size = argc > 1 ? atoi(argv[1]) : 400;
  const auto tw = mfcc::twiddle512();
  double worst = 0;
  for (int trial = 0; trial < 8; ++trial) {
    std::vector<float> x(512, 0.f);
    srand(1234 + trial);
    for (int n = 0; n < frame_size; ++n) x[n] = (float)((rand() % 65536) - 32768) * (trial == 7 ? 1.0f : 0.25f);
    // ---- stage 1 : lane q = n2
    std::vector<float> T(512, 0.f);  // 256 complex, exchange buffer
    for (int q = 0; q < 16; ++q) {
      mfcc::cf v[16];
      for (int n1 = 0; n1 < 16; ++n1) {
        const int n = 16 * n1 + q;
        v[n1] = mfcc::mk(x[2 * n], x[2 * n + 1]);
      }
      mfcc::fft16(v);
      for (int k1 = 0; k1 < 16; ++k1) {
        const int m = (2 * q * k1) & 511;  // W256^(q k1) = W512^(2 q k1)
        if (k1) v[k1] = mfcc::cmul(v[k1], mfcc::mk(tw[2 * m], tw[2 * m + 1]));
        const int p = mfcc::xpose_index(k1, q);
        T[2 * p] = v[k1].x;
        T[2 * p + 1] = v[k1].y;
      }
    }
    // ---- stage 2 : lane q = k1
    mfcc::cf Z[16][16];
    for (int q = 0; q < 16; ++q) {
      for (int n2 = 0; n2 < 16; ++n2) {
        const int p = mfcc::xpose_index(q, n2);
        Z[q][n2] = mfcc::mk(T[2 * p], T[2 * p + 1]);
      }
      mfcc::fft16(Z[q]);
    }
    // ---- untangle
    std::vector<float> P(257, -1.f);
    for (int q = 0; q < 16; ++q) {
      const int pl = mfcc::partner_lane(q);
      for (int s = 0; s < 8; ++s) {
        const int k = q + 16 * s;
        mfcc::cf b = Z[pl][15 - s];
        if (q == 0) b = Z[0][(16 - s) & 15];
        float lo, hi;
        if ((trial & 1) == 0) mfcc::untangle_power_tc(Z[q][s], b, mfcc::mk(tw[2 * (512 + k)], tw[2 * (512 + k) + 1]), lo, hi);
        else mfcc::untangle_power(Z[q][s], b, mfcc::mk(tw[2 * k], tw[2 * k + 1]), lo, hi);
        if (P[k] != -1.f || P[256 - k] != -1.f) { printf("bin written twice: %d\n", k); return 2; }
        P[k] = lo;
        P[256 - k] = hi;
      }
      if (q == 0) {
        if (P[128] != -1.f) { printf("bin 128 written twice\n"); return 2; }
        P[128] = 4.f * mfcc::norm2(Z[0][8]);
      }
    }
    // ---- reference
    const double pi = 3.14159265358979323846;
    double rms = 0;
    std::vector<double> ref(257);
    for (int k = 0; k <= 256; ++k) {
      double re = 0, im = 0;
      for (int n = 0; n < frame_size; ++n) {
        re += x[n] * std::cos(2 * pi * k * n / 512.0);
        im -= x[n] * std::sin(2 * pi * k * n / 512.0);
      }
      ref[k] = 4 * (re * re + im * im);
      rms += ref[k];
    }
    rms /= 257;
    for (int k = 0; k <= 256; ++k) {
      if (P[k] == -1.f) { printf("bin %d never written\n", k); return 2; }
      const double e = std::fabs(P[k] - ref[k]) / rms;
      if (e > worst) worst = e;
    }
  }
  printf("max_err_rel_rms %.3e\n", worst);
  return worst < 2e-6 ? 0 : 1;
}
