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
// (mfcc-self5.py:96).
//
// A complex value is one 64-bit register pair and every butterfly is written with the
// packed-f32 instructions of gfx950 (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32): on this
// chip a packed instruction costs about 1.25x a scalar one and does twice the work.  Swaps
// and sign flips (multiplication by -i, conjugation) ride on the op_sel / neg modifiers of
// the packed instruction that consumes them, which hipcc does not fold by itself, hence
// the inline assembly.  On the host the same functions are plain C++ so that
// tests/host_sim exercises the identical index logic lane by lane.
#pragma once

#if defined(__HIPCC__)
#define MFCC_HD __host__ __device__ __forceinline__
#else
#define MFCC_HD inline
#endif

namespace mfcc {

#if defined(__clang__)
typedef float cf __attribute__((ext_vector_type(2)));
MFCC_HD cf mk(float x, float y) { cf r; r.x = x; r.y = y; return r; }
#else
struct cf {
  float x, y;
};
MFCC_HD cf mk(float x, float y) { return cf{x, y}; }
MFCC_HD cf operator+(cf a, cf b) { return cf{a.x + b.x, a.y + b.y}; }
MFCC_HD cf operator-(cf a, cf b) { return cf{a.x - b.x, a.y - b.y}; }
MFCC_HD cf operator*(cf a, cf b) { return cf{a.x * b.x, a.y * b.y}; }
#endif

constexpr float kC1 = 0.92387953251128674f;  // cos(pi/8)
constexpr float kS1 = 0.38268343236508977f;  // sin(pi/8)
constexpr float kR2 = 0.70710678118654752f;  // sqrt(1/2)
constexpr float kT1 = 0.41421356237309505f;  // tan(pi/8) = kS1 / kC1
constexpr float kT3 = 2.41421356237309505f;  // tan(3 pi/8) = kC1 / kS1

#if defined(__HIP_DEVICE_COMPILE__)
// a + (-i) b = (a.x + b.y, a.y - b.x)
__device__ __forceinline__ cf add_mi(cf a, cf b) {
  cf d;
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
// a - (-i) b = (a.x - b.y, a.y + b.x)
__device__ __forceinline__ cf sub_mi(cf a, cf b) {
  cf d;
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
// a + conj(b)
__device__ __forceinline__ cf add_conj(cf a, cf b) {
  cf d;
  asm("v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
// a - conj(b)
__device__ __forceinline__ cf sub_conj(cf a, cf b) {
  cf d;
  asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
// (-i) a = (a.y, -a.x)
__device__ __forceinline__ cf mul_mi(cf a) {
  cf d;
  asm("v_pk_mul_f32 %0, %1, 1.0 op_sel:[1,0] op_sel_hi:[0,0] neg_hi:[1,0]" : "=v"(d) : "v"(a));
  return d;
}
// a * w, w in vector registers (lane-dependent twiddle).  Both instructions sit in one asm
// statement: the compiler assumes that an asm statement's result may need a wait state before
// its first use and pads with s_nop when the use comes next; inside the statement the hardware's
// own interlock orders the two.
__device__ __forceinline__ cf cmul(cf a, cf w) {
  cf t, d;
  asm("v_pk_mul_f32 %1, %2, %3 op_sel:[0,0] op_sel_hi:[0,1]\n\t"
      "v_pk_fma_f32 %0, %2, %3, %1 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]"
      : "=v"(d), "=&v"(t) : "v"(a), "v"(w));
  return d;
}
// a * w, w wave-uniform (scalar register pair)
__device__ __forceinline__ cf cmulk(cf a, cf w) {
  cf t, d;
  asm("v_pk_mul_f32 %1, %2, %3 op_sel:[0,0] op_sel_hi:[0,1]\n\t"
      "v_pk_fma_f32 %0, %2, %3, %1 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]"
      : "=v"(d), "=&v"(t) : "v"(a), "s"(w));
  return d;
}
// two products at once, interleaved so that no instruction uses the result of the one before it
__device__ __forceinline__ void cmul2(cf& a0, cf w0, cf& a1, cf w1) {
  cf t0, t1;
  asm("v_pk_mul_f32 %2, %0, %4 op_sel:[0,0] op_sel_hi:[0,1]\n\t"
      "v_pk_mul_f32 %3, %1, %5 op_sel:[0,0] op_sel_hi:[0,1]\n\t"
      "v_pk_fma_f32 %0, %0, %4, %2 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]\n\t"
      "v_pk_fma_f32 %1, %1, %5, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]"
      : "+v"(a0), "+v"(a1), "=&v"(t0), "=&v"(t1) : "v"(w0), "v"(w1));
}
// (a.x + b.y, a.x - b.y) and (a.y - b.x, a.y + b.x): the x parts, resp. y parts, of a - i b and a + i b
__device__ __forceinline__ cf pair_x(cf a, cf b) {
  cf d;
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
__device__ __forceinline__ cf pair_y(cf a, cf b) {
  cf d;
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
// a + k p, a - k p: k wave-uniform, the same factor for both halves (a scalar register pair (k, k))
__device__ __forceinline__ cf fmak(cf p, cf k, cf a) {
  cf d;
  asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(p), "s"(k), "v"(a));
  return d;
}
__device__ __forceinline__ cf fmak_neg(cf p, cf k, cf a) {
  cf d;
  asm("v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[0,1,0] neg_hi:[0,1,0]" : "=v"(d) : "v"(p), "s"(k), "v"(a));
  return d;
}
// a + k (-i p) = (a.x + k p.y, a.y - k p.x)
__device__ __forceinline__ cf fmak_mi(cf p, cf k, cf a) {
  cf d;
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[0,1,1] neg_hi:[0,1,0]" : "=v"(d) : "v"(p), "s"(k), "v"(a));
  return d;
}
// a + k (i p) = (a.x - k p.y, a.y + k p.x)
__device__ __forceinline__ cf fmak_pi(cf p, cf k, cf a) {
  cf d;
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]" : "=v"(d) : "v"(p), "s"(k), "v"(a));
  return d;
}
// One untangle step with the twiddle W512^k = c (1 - i t) given as tc = (t, c) = (tan, cos)(2 pi k / 512):
// see untangle_power_tc below.  C' = B (1 - i t); X = (A.x + c C'.y, A.x - c C'.y); Y = (A.y - c C'.x, A.y + c C'.x)
__device__ __forceinline__ cf unt_rot(cf b, cf tc) {
  cf d;
  asm("v_pk_fma_f32 %0, %1, %2, %1 op_sel:[1,0,0] op_sel_hi:[0,0,1] neg_hi:[0,1,0]" : "=v"(d) : "v"(b), "v"(tc));
  return d;
}
__device__ __forceinline__ cf unt_x(cf a, cf c, cf tc) {
  cf d;
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,1,0] neg_hi:[0,1,0]" : "=v"(d) : "v"(c), "v"(tc), "v"(a));
  return d;
}
__device__ __forceinline__ cf unt_y(cf a, cf c, cf tc) {
  cf d;
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,1] op_sel_hi:[0,1,1] neg_lo:[0,1,0]" : "=v"(d) : "v"(c), "v"(tc), "v"(a));
  return d;
}
// |a|^2
__device__ __forceinline__ float norm2(cf a) {
  const cf t = a * a;
  return t.x + t.y;
}
#else
MFCC_HD cf add_mi(cf a, cf b) { return mk(a.x + b.y, a.y - b.x); }
MFCC_HD cf sub_mi(cf a, cf b) { return mk(a.x - b.y, a.y + b.x); }
MFCC_HD cf add_conj(cf a, cf b) { return mk(a.x + b.x, a.y - b.y); }
MFCC_HD cf sub_conj(cf a, cf b) { return mk(a.x - b.x, a.y + b.y); }
MFCC_HD cf mul_mi(cf a) { return mk(a.y, -a.x); }
MFCC_HD cf cmul(cf a, cf w) { return mk(a.x * w.x - a.y * w.y, a.x * w.y + a.y * w.x); }
MFCC_HD cf cmulk(cf a, cf w) { return cmul(a, w); }
MFCC_HD void cmul2(cf& a0, cf w0, cf& a1, cf w1) { a0 = cmul(a0, w0); a1 = cmul(a1, w1); }
MFCC_HD cf pair_x(cf a, cf b) { return mk(a.x + b.y, a.x - b.y); }
MFCC_HD cf pair_y(cf a, cf b) { return mk(a.y - b.x, a.y + b.x); }
MFCC_HD cf fmak(cf p, cf k, cf a) { return mk(a.x + k.x * p.x, a.y + k.y * p.y); }
MFCC_HD cf fmak_neg(cf p, cf k, cf a) { return mk(a.x - k.x * p.x, a.y - k.y * p.y); }
MFCC_HD cf fmak_mi(cf p, cf k, cf a) { return mk(a.x + k.x * p.y, a.y - k.y * p.x); }
MFCC_HD cf fmak_pi(cf p, cf k, cf a) { return mk(a.x - k.x * p.y, a.y + k.y * p.x); }
MFCC_HD cf unt_rot(cf b, cf tc) { return mk(b.x + tc.x * b.y, b.y - tc.x * b.x); }
MFCC_HD cf unt_x(cf a, cf c, cf tc) { return mk(a.x + tc.y * c.y, a.x - tc.y * c.y); }
MFCC_HD cf unt_y(cf a, cf c, cf tc) { return mk(a.y - tc.y * c.x, a.y + tc.y * c.x); }
MFCC_HD float norm2(cf a) { return a.x * a.x + a.y * a.y; }
#endif

// Forward radix-4 butterfly, in place:
//   a0' = a0+a1+a2+a3, a1' = a0 - i a1 - a2 + i a3, a2' = a0-a1+a2-a3, a3' = a0 + i a1 - a2 - i a3
MFCC_HD void radix4(cf& a0, cf& a1, cf& a2, cf& a3) {
  const cf t0 = a0 + a2, t1 = a0 - a2, t2 = a1 + a3, t3 = a1 - a3;
  a0 = t0 + t2;
  a2 = t0 - t2;
  a1 = add_mi(t1, t3);
  a3 = sub_mi(t1, t3);
}

// 16-point forward DFT, natural order in and out, all indices compile-time.
// x[n], n = 4a + b;  X[c + 4d] = sum_b W4^(bd) W16^(bc) sum_a x[4a+b] W4^(ac).
// Two halves: the radix-4 butterflies over a (fft16_a), then twiddles and butterflies over b (fft16_bc).
MFCC_HD void fft16_a(cf (&x)[16]) {
#pragma unroll
  for (int b = 0; b < 4; ++b) radix4(x[b], x[4 + b], x[8 + b], x[12 + b]);
}
// First pass for a frame whose inputs x[NZ..15] are zero (a frame of frame_size real samples
// fills complex points 0 .. frame_size/2 - 1, i.e. registers n1 < ceil(frame_size / 32)): the
// butterflies lose the additions whose operand is a known zero.
template <int NZ>
MFCC_HD void fft16_a_pruned(cf (&x)[16]) {
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    cf &a0 = x[b], &a1 = x[4 + b], &a2 = x[8 + b], &a3 = x[12 + b];
    if (12 + b < NZ) {
      radix4(a0, a1, a2, a3);
    } else if (8 + b < NZ) {        // a3 = 0
      const cf t0 = a0 + a2, t1 = a0 - a2, u = a1;
      a0 = t0 + u;
      a2 = t0 - u;
      a1 = add_mi(t1, u);
      a3 = sub_mi(t1, u);
    } else if (4 + b < NZ) {        // a2 = a3 = 0
      const cf t = a0, u = a1;
      a0 = t + u;
      a2 = t - u;
      a1 = add_mi(t, u);
      a3 = sub_mi(t, u);
    } else {                        // only a0
      a1 = a0;
      a2 = a0;
      a3 = a0;
    }
  }
}
// Second half of the 16-point transform: after the first pass x[4c + b] holds Y_b[c]; what follows is the
// twiddle W16^(bc) and a radix-4 butterfly over b for every c.  A twiddle W16^m = C_m (1 - i T_m) (C = cos,
// T = tan of pi m / 8) is applied as the rotation u (1 - i T) -- one packed fma -- and its scale C_m rides on
// the packed fma that consumes the value in the butterfly, so a twiddled input costs one instruction instead
// of the two of a complex product; for m = 2, 4, 6 (T = +-1 or a pure -i) the rotation is a packed add:
//   c = 1:  W^1 = C1 (1 - i T1), W^2 = R2 (1 - i),  W^3 = S1 (1 - i T3)      S1 / C1 = T1
//   c = 2:  W^2 = R2 (1 - i),    W^4 = -i,          W^6 = -R2 (1 + i)
//   c = 3:  W^3 = S1 (1 - i T3), W^6 = -R2 (1 + i), W^9 = -C1 (1 - i T1)     C1 / S1 = T3
// 40 packed instructions against 47 for nine complex products and four butterflies.
MFCC_HD void fft16_bc(cf (&x)[16]) {
  const cf c1 = mk(kC1, kC1), s1 = mk(kS1, kS1), r2 = mk(kR2, kR2), t1 = mk(kT1, kT1), t3 = mk(kT3, kT3);
  radix4(x[0], x[1], x[2], x[3]);
  {  // c = 1
    const cf p1 = fmak_mi(x[5], t1, x[5]), p2 = add_mi(x[6], x[6]), p3 = fmak_mi(x[7], t3, x[7]);
    const cf u0 = fmak(p2, r2, x[4]), u1 = fmak_neg(p2, r2, x[4]);
    const cf u2 = fmak(p3, t1, p1), u3 = fmak_neg(p3, t1, p1);  // (a1 + a3) / C1, (a1 - a3) / C1
    x[4] = fmak(u2, c1, u0);
    x[6] = fmak_neg(u2, c1, u0);
    x[5] = fmak_mi(u3, c1, u1);
    x[7] = fmak_pi(u3, c1, u1);
  }
  {  // c = 2
    const cf p1 = add_mi(x[9], x[9]), p3 = sub_mi(x[11], x[11]);
    const cf u0 = add_mi(x[8], x[10]), u1 = sub_mi(x[8], x[10]);
    const cf u2 = p1 - p3, u3 = p1 + p3;  // (a1 + a3) / R2, (a1 - a3) / R2: a3 = -R2 p3
    x[8] = fmak(u2, r2, u0);
    x[10] = fmak_neg(u2, r2, u0);
    x[9] = fmak_mi(u3, r2, u1);
    x[11] = fmak_pi(u3, r2, u1);
  }
  {  // c = 3
    const cf p1 = fmak_mi(x[13], t3, x[13]), p2 = sub_mi(x[14], x[14]), p3 = fmak_mi(x[15], t1, x[15]);
    const cf u0 = fmak_neg(p2, r2, x[12]), u1 = fmak(p2, r2, x[12]);  // a2 = -R2 p2
    const cf u2 = fmak_neg(p3, t3, p1), u3 = fmak(p3, t3, p1);  // (a1 + a3) / S1, (a1 - a3) / S1: a3 = -C1 p3
    x[12] = fmak(u2, s1, u0);
    x[14] = fmak_neg(u2, s1, u0);
    x[13] = fmak_mi(u3, s1, u1);
    x[15] = fmak_pi(u3, s1, u1);
  }
  // x[4c + d] now holds X[c + 4d]: transpose the 4x4 index grid (register renaming only)
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int d = c + 1; d < 4; ++d) {
      const cf t = x[4 * c + d];
      x[4 * c + d] = x[4 * d + c];
      x[4 * d + c] = t;
    }
}
MFCC_HD void fft16(cf (&x)[16]) {
  fft16_a(x);
  fft16_bc(x);
}
template <int NZ>
MFCC_HD void fft16_pruned(cf (&x)[16]) {
  fft16_a_pruned<NZ>(x);
  fft16_bc(x);
}

// Position (in complex elements) of Y[k1][n2] inside one frame's 256-element exchange
// buffer.  Row k1 is 16 elements (128 B); XOR-ing the column with (k1 & 14) makes both the
// 8-byte column writes of stage 1 and the 16-byte row reads of stage 2 bank-conflict free.
MFCC_HD int xpose_index(int k1, int n2) { return k1 * 16 + (n2 ^ (k1 & 14)); }

// Lane that holds Z[256 - k] for the k values of lane q.
MFCC_HD int partner_lane(int q) { return (16 - q) & 15; }

// One untangle step.  a = Z[k], b = Z[256-k], w = W512^k.
// Returns 4|X[k]|^2 in p_lo and 4|X[256-k]|^2 in p_hi:
//   2 X[k]       = A - i W B,   2 conj X[256-k] = A + i W B,
//   A = Z[k] + conj Z[256-k],   B = Z[k] - conj Z[256-k].
MFCC_HD void untangle_power(cf a, cf b, cf w, float& p_lo, float& p_hi) {
  const cf A = add_conj(a, b);
  const cf B = sub_conj(a, b);
  const cf C = cmul(B, w);
  // U = A - iC, V = A + iC held as X = (U.x, V.x), Y = (U.y, V.y): both squared moduli come out
  // of one packed multiply and one packed fma
  const cf X = pair_x(A, C), Y = pair_y(A, C);
#if defined(__HIP_DEVICE_COMPILE__)
  const cf p = __builtin_elementwise_fma(Y, Y, X * X);
#else
  const cf p = mk(X.x * X.x + Y.x * Y.x, X.y * X.y + Y.y * Y.y);
#endif
  p_lo = p.x;
  p_hi = p.y;
}


// The same step with the twiddle W512^k = c (1 - i t) given as tc = (t, c) = (tan, cos)(2 pi k / 512), k < 128:
// C' = B (1 - i t) is one packed fma and the factor c rides on the two fmas that form X and Y, seven packed
// instructions instead of eight.  (tan stays below 82 for k <= 127; the products c * (t * B) round like W * B.)
MFCC_HD void untangle_power_tc(cf a, cf b, cf tc, float& p_lo, float& p_hi) {
  const cf A = add_conj(a, b);
  const cf B = sub_conj(a, b);
  const cf C = unt_rot(B, tc);
  const cf X = unt_x(A, C, tc), Y = unt_y(A, C, tc);
#if defined(__HIP_DEVICE_COMPILE__)
  const cf p = __builtin_elementwise_fma(Y, Y, X * X);
#else
  const cf p = mk(X.x * X.x + Y.x * Y.x, X.y * X.y + Y.y * Y.y);
#endif
  p_lo = p.x;
  p_hi = p.y;
}

}  // namespace mfcc
