// Host-side (double precision) construction of the constant tables of a plan.
// Each routine states the reference lines whose constants it reproduces.
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <vector>

namespace mfcc {

// frame_size = int(fs * (frame_time / 1000)), frame_step likewise, num_fft = next
// power of two >= frame_size.  mfcc-self5.py:75-84.
inline void frame_geometry(double fs, double frame_time, double step_time, int32_t* size, int32_t* step,
                           int32_t* nfft) {
  *size = (int32_t)(fs * (frame_time * (1.0 / 1000)));
  *step = (int32_t)(fs * (step_time * (1.0 / 1000)));
  int32_t n = 1;
  while (n < *size) n *= 2;
  *nfft = n;
}

// Smallest pos with pos*step + size >= L, frames = pos + 1.  mfcc-self5.py:85-92.
inline int64_t frames_for_length(int64_t n, int32_t size, int32_t step) {
  if (n <= size) return 1;
  return 1 + (n - size + step - 1) / step;
}

// Triangular mel bank [numbins][num_fft/2+1], HTK mel scale, peaks at
// floor((num_fft+1) * hz / fs), no area normalisation.  mfcc-self5.py:102-117.
// highfreq <= 0 selects the reference's fixed upper edge fs / 2 (:103).
inline std::vector<double> melbank(double fs, int numbins, double lowfreq, int num_fft, double highfreq = 0) {
  const int nb = num_fft / 2 + 1;
  if (!(highfreq > 0)) highfreq = fs / 2;
  const double mel_lo = 2595 * std::log10(1 + lowfreq / 700.0);
  const double mel_hi = 2595 * std::log10(1 + highfreq / 700.0);
  std::vector<double> edge(numbins + 2);
  for (int i = 0; i < numbins + 2; ++i) {
    // numpy.linspace(start, stop, n): start + i * step, last point exactly stop
    const double step = (mel_hi - mel_lo) / (numbins + 1);
    const double m = (i == numbins + 1) ? mel_hi : mel_lo + i * step;
    const double hz = 700 * (std::pow(10.0, m / 2595.0) - 1);
    edge[i] = std::floor((num_fft + 1) * hz / fs);
  }
  std::vector<double> bank((size_t)numbins * nb, 0.0);
  for (int j = 0; j < numbins; ++j) {
    const double lo = edge[j], mid = edge[j + 1], hi = edge[j + 2];
    for (int i = (int)lo; i < (int)mid && i < nb; ++i) bank[(size_t)j * nb + i] = (i - lo) / (mid - lo);
    for (int i = (int)mid; i < (int)hi && i < nb; ++i) bank[(size_t)j * nb + i] = (hi - i) / (hi - mid);
  }
  return bank;
}

// DCT-II with norm='ortho' truncated to numcep columns (scipy.fftpack.dct as called at
// mfcc-self5.py:140), times the sinusoidal lifter 1 + (L/2) sin(pi k / L) of
// mfcc-self5.py:149-151 when lift > 0.  Layout [nmel][numcep].
inline std::vector<double> dct_lifter(int nmel, int numcep, double lift, bool with_lifter) {
  const double pi = 3.14159265358979323846;
  std::vector<double> w((size_t)nmel * numcep);
  for (int n = 0; n < nmel; ++n)
    for (int k = 0; k < numcep; ++k) {
      const double s = (k == 0) ? std::sqrt(1.0 / nmel) : std::sqrt(2.0 / nmel);
      double v = s * std::cos(pi * k * (2 * n + 1) / (2.0 * nmel));
      if (with_lifter && lift > 0) v *= 1 + (lift / 2.0) * std::sin(pi * k / lift);
      w[(size_t)n * numcep + k] = v;
    }
  return w;
}

inline std::vector<double> lifter(int ncoeff, double lift) {
  const double pi = 3.14159265358979323846;
  std::vector<double> w(ncoeff, 1.0);
  if (lift > 0)
    for (int k = 0; k < ncoeff; ++k) w[k] = 1 + (lift / 2.0) * std::sin(pi * k / lift);
  return w;
}

// np.hamming(M)[n] = 0.54 - 0.46 cos(2 pi n / (M-1))  (returned, unused, by winfft: mfcc-self5.py:78).
inline std::vector<double> hamming(int m) {
  const double pi = 3.14159265358979323846;
  std::vector<double> w(m, 1.0);
  if (m > 1)
    for (int n = 0; n < m; ++n) w[n] = 0.54 - 0.46 * std::cos(2 * pi * n / (m - 1));
  return w;
}

// W_n^k = exp(-2 pi i k / n), k = 0 .. n/2 - 1, interleaved (re, im) as float.
inline std::vector<float> twiddle_half(int n) {
  const double pi = 3.14159265358979323846;
  std::vector<float> t((size_t)std::max(1, n / 2) * 2, 0.f);
  for (int k = 0; k < n / 2; ++k) {
    t[2 * k] = (float)std::cos(2 * pi * k / n);
    t[2 * k + 1] = (float)(-std::sin(2 * pi * k / n));
  }
  if (n < 2) t[0] = 1.f;
  return t;
}

// W512^m = exp(-2 pi i m / 512), m = 0..511, interleaved (re, im) as float; behind them (entries
// 512 .. 639) the untangle twiddles of the 16-lane transform in the form W512^k = c (1 - i t):
// (t, c) = (tan, cos)(2 pi k / 512) for k = 0..127.
constexpr int kTw512Entries = 512 + 128;
inline std::vector<float> twiddle512() {
  const double pi = 3.14159265358979323846;
  std::vector<float> t(2 * kTw512Entries);
  for (int m = 0; m < 512; ++m) {
    t[2 * m] = (float)std::cos(2 * pi * m / 512.0);
    t[2 * m + 1] = (float)(-std::sin(2 * pi * m / 512.0));
  }
  for (int k = 0; k < 128; ++k) {
    t[2 * (512 + k)] = (float)std::tan(2 * pi * k / 512.0);
    t[2 * (512 + k) + 1] = (float)std::cos(2 * pi * k / 512.0);
  }
  return t;
}

}  // namespace mfcc
