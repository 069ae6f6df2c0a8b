// Lane-level pieces of the 512-point real FFT used by the fused MFCC kernel.
//
// A frame of up to 512 real samples x[0..511] (zero beyond frame_size) is packed
// as 256 complex points z[n] = x[2n] + i x[2n+1] and transformed by a 16 x 16
// Cooley-Tukey split carried by 16 lanes:
//
//   stage 1  lane q = n2 :  Y[k1] = W256^(q k1) * sum_n1 z[16 n1 + q] W16^(n1 k1)
//   exchange through LDS  :  lane q = k1 receives Y[k1][n2 = 0..15]
//   stage 2  lane q = k1 :  Z[k1 + 16 k2] = sum_n2 Y[k1][n2] W16^(n2 k2)
//   untangle             :  X[k], X[256-k] from Z[k], Z[256-k]  (k = q + 16 k2, k2 < 8)
//
// This replaces the per-frame np.fft.rfft(chunk, 512) of the reference
// (mfcc-self5.py:96).  Everything here is plain C++ so that the same code is
// exercised lane by lane on the host by tests/host_sim.
#pragma once

#if defined(__HIPCC__)
#define MFCC_HD __host__ __device__ __forceinline__
#else
#define MFCC_HD inline
#endif

namespace mfcc {

constexpr float kC1 = 0.92387953251128674f;  // cos(pi/8)
constexpr float kS1 = 0.38268343236508977f;  // sin(pi/8)
constexpr float kR2 = 0.70710678118654752f;  // sqrt(1/2)

// Forward radix-4 butterfly, in place:
//   a0' = a0+a1+a2+a3, a1' = a0 - i a1 - a2 + i a3, a2' = a0-a1+a2-a3, a3' = a0 + i a1 - a2 - i a3
MFCC_HD void radix4(float& r0, float& i0, float& r1, float& i1, float& r2, float& i2, float& r3, float& i3) {
  const float t0r = r0 + r2, t0i = i0 + i2;
  const float t1r = r0 - r2, t1i = i0 - i2;
  const float t2r = r1 + r3, t2i = i1 + i3;
  const float t3r = r1 - r3, t3i = i1 - i3;
  r0 = t0r + t2r; i0 = t0i + t2i;
  r2 = t0r - t2r; i2 = t0i - t2i;
  r1 = t1r + t3i; i1 = t1i - t3r;
  r3 = t1r - t3i; i3 = t1i + t3r;
}

MFCC_HD void cmul(float& r, float& i, const float wr, const float wi) {
  const float nr = r * wr - i * wi;
  const float ni = r * wi + i * wr;
  r = nr; i = ni;
}

// 16-point forward DFT, natural order in and out, all indices compile-time.
// x[n], n = 4a + b;  X[c + 4d] = sum_b W4^(bd) W16^(bc) sum_a x[4a+b] W4^(ac).
MFCC_HD void fft16(float (&xr)[16], float (&xi)[16]) {
#pragma unroll
  for (int b = 0; b < 4; ++b)
    radix4(xr[b], xi[b], xr[4 + b], xi[4 + b], xr[8 + b], xi[8 + b], xr[12 + b], xi[12 + b]);
  // after the first pass xr[4c + b] holds Y_b[c]; twiddle by W16^(bc), W16^m = (cos, -sin)(2 pi m / 16)
  cmul(xr[4 + 1], xi[4 + 1], kC1, -kS1);    // b=1 c=1  m=1
  cmul(xr[8 + 1], xi[8 + 1], kR2, -kR2);    // b=1 c=2  m=2
  cmul(xr[12 + 1], xi[12 + 1], kS1, -kC1);  // b=1 c=3  m=3
  cmul(xr[4 + 2], xi[4 + 2], kR2, -kR2);    // b=2 c=1  m=2
  {                                         // b=2 c=2  m=4 : multiply by -i
    const float t = xr[8 + 2]; xr[8 + 2] = xi[8 + 2]; xi[8 + 2] = -t;
  }
  cmul(xr[12 + 2], xi[12 + 2], -kR2, -kR2);  // b=2 c=3  m=6
  cmul(xr[4 + 3], xi[4 + 3], kS1, -kC1);     // b=3 c=1  m=3
  cmul(xr[8 + 3], xi[8 + 3], -kR2, -kR2);    // b=3 c=2  m=6
  cmul(xr[12 + 3], xi[12 + 3], -kC1, kS1);   // b=3 c=3  m=9
#pragma unroll
  for (int c = 0; c < 4; ++c)
    radix4(xr[4 * c], xi[4 * c], xr[4 * c + 1], xi[4 * c + 1], xr[4 * c + 2], xi[4 * c + 2], xr[4 * c + 3],
           xi[4 * c + 3]);
  // xr[4c + d] now holds X[c + 4d]: transpose the 4x4 index grid (register renaming only)
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int d = c + 1; d < 4; ++d) {
      float t = xr[4 * c + d]; xr[4 * c + d] = xr[4 * d + c]; xr[4 * d + c] = t;
      t = xi[4 * c + d]; xi[4 * c + d] = xi[4 * d + c]; xi[4 * d + c] = t;
    }
}

// Position (in complex elements) of Y[k1][n2] inside one frame's 256-element exchange
// buffer.  Row k1 is 16 elements (128 B); XOR-ing the column with (k1 & 14) makes both the
// 8-byte column writes of stage 1 and the 16-byte row reads of stage 2 bank-conflict free.
MFCC_HD int xpose_index(int k1, int n2) { return k1 * 16 + (n2 ^ (k1 & 14)); }

// Lane that holds Z[256 - k] for the k values of lane q.
MFCC_HD int partner_lane(int q) { return (16 - q) & 15; }

// One untangle step.  (ar, ai) = Z[k], (br, bi) = Z[256-k], (wr, wi) = W512^k.
// Returns 4|X[k]|^2 in p_lo and 4|X[256-k]|^2 in p_hi:
//   2 X[k]       = A - i W B,   2 conj X[256-k] = A + i W B,
//   A = Z[k] + conj Z[256-k],   B = Z[k] - conj Z[256-k].
MFCC_HD void untangle_power(float ar, float ai, float br, float bi, float wr, float wi, float& p_lo, float& p_hi) {
  const float Ar = ar + br, Ai = ai - bi;
  const float Br = ar - br, Bi = ai + bi;
  const float Cr = wr * Br - wi * Bi;
  const float Ci = wr * Bi + wi * Br;
  const float x0r = Ar + Ci, x0i = Ai - Cr;
  const float x1r = Ar - Ci, x1i = Ai + Cr;
  p_lo = x0r * x0r + x0i * x0i;
  p_hi = x1r * x1r + x1i * x1i;
}

}  // namespace mfcc
