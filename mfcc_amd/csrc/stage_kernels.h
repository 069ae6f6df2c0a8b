// One kernel per reference stage, for callers that drive the pipeline function by
// function (the reference's main(), mfcc-self5.py:211-229) and for stage-level parity
// tests.  These are simple bandwidth-shaped kernels; the fused kernel is the hot path.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fused_kernel.h"  // Tile, kTileSums

namespace mfcc {

constexpr float kEps64s = 2.220446049250313e-16f;

// y[0] = x[0]; y[n] = x[n] - c x[n-1].  mfcc-self5.py:66-68
template <typename T>
__global__ void preemph_kernel(const T* __restrict__ x, int64_t n, float c_hi, float c_lo, float* __restrict__ y) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float x0 = (float)x[i];
  if (i == 0) { y[0] = x0; return; }
  const float x1 = (float)x[i - 1];
  if (sizeof(T) == 2) {
    y[i] = fmaf(-c_lo, x1, fmaf(-c_hi, x1, x0));
  } else {
    y[i] = __fsub_rn(x0, __fmul_rn(c_hi, x1));  // numpy's float32 arithmetic for float32 input
  }
}

// One workgroup (64 threads) per frame: ln(spec[f] . bank[j]) for every filter j, and
// ln(sum spec[f]) floored at eps.  mfcc-self5.py:119-134
__global__ void melspec_kernel(const float* __restrict__ spec, int n_bins, const float* __restrict__ bank, int nmel,
                               int floor_mel, float* __restrict__ logmel, float* __restrict__ loge) {
  const int64_t f = blockIdx.x;
  const float* row = spec + f * n_bins;
  const int lane = threadIdx.x;
  for (int j = 0; j <= nmel; ++j) {
    float acc = 0.f;
    if (j < nmel) {
      const float* w = bank + (int64_t)j * n_bins;
      for (int k = lane; k < n_bins; k += 64) acc = fmaf(row[k], w[k], acc);
    } else {
      for (int k = lane; k < n_bins; k += 64) acc += row[k];
    }
    for (int m = 32; m >= 1; m >>= 1) acc += __shfl_xor(acc, m, 64);
    if (lane == 0) {
      if (j < nmel) {
        if (floor_mel && acc == 0.f) acc = kEps64s;
        logmel[f * nmel + j] = logf(acc);
      } else {
        if (acc == 0.f) acc = kEps64s;
        loge[f] = logf(acc);
      }
    }
  }
}

// DCT-II, norm='ortho', first numcep coefficients (scipy.fftpack.dct as called at
// mfcc-self5.py:140): c_k = s_k sum_n x_n cos(pi k (2n+1) / (2N)).  The cosines are
// evaluated in double so that the float32 weights equal the rounded exact values.
__global__ void dct_kernel(const float* __restrict__ in, int64_t n_rows, int nmel, int numcep, float* __restrict__ out) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n_rows * numcep) return;
  const int64_t f = idx / numcep;
  const int k = (int)(idx - f * numcep);
  const double s = k == 0 ? sqrt(1.0 / nmel) : sqrt(2.0 / nmel);
  float acc = 0.f;
  for (int n = 0; n < nmel; ++n) {
    const float w = (float)(s * cospi((double)(k * (2 * n + 1)) / (2.0 * nmel)));
    acc = fmaf(in[f * nmel + n], w, acc);
  }
  out[idx] = acc;
}

// c_n <- c_n (1 + (L/2) sin(pi n / L)) for L > 0, identity otherwise.  mfcc-self5.py:144-156
__global__ void lifter_kernel(const float* __restrict__ in, int64_t n_rows, int ncol, double lift, float* __restrict__ out) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n_rows * ncol) return;
  const int n = (int)(idx % ncol);
  const float w = lift > 0 ? (float)(1.0 + (lift / 2.0) * sinpi((double)n / lift)) : 1.f;
  out[idx] = in[idx] * w;
}

// d[t] = sum_{k=-N..N} k x[clamp(t+k)] / (2 sum k^2).  mfcc-self5.py:169-179
__global__ void delta_kernel(const float* __restrict__ x, int64_t n_rows, int ncol, int pad_fac, float inv_denom,
                             float* __restrict__ out) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n_rows * ncol) return;
  const int64_t t = idx / ncol;
  const int c = (int)(idx - t * ncol);
  float acc = 0.f;
  for (int k = -pad_fac; k <= pad_fac; ++k) {
    int64_t tt = t + k;
    tt = tt < 0 ? 0 : (tt > n_rows - 1 ? n_rows - 1 : tt);
    acc = fmaf((float)k, x[tt * ncol + c], acc);
  }
  out[idx] = acc * inv_denom;
}

__global__ void copy_cols_kernel(const float* __restrict__ src, int64_t src_ld, float* __restrict__ dst,
                                 int64_t dst_ld, int64_t n_rows, int ncol) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n_rows * ncol) return;
  const int64_t r = idx / ncol;
  const int c = (int)(idx - r * ncol);
  dst[r * dst_ld + c] = src[r * src_ld + c];
}

// Centred FIR per utterance: out[n] = sum_k taps[k] * x[n + delay - k], zero outside the
// utterance (filter() + group-delay shift of spec_anal, anal_recog_nn_liste.m:511-516).  A block
// covers 1024 consecutive samples of the packed buffer; its first utterance is found by bisection
// once, threads then step forward over the (rare) boundaries inside the block.
struct FirTaps {
  float w[32];
  int32_t n, delay;
};
template <typename T>
__global__ void __launch_bounds__(256) fir_kernel(const T* __restrict__ x, const int64_t* __restrict__ sample_off, int n_utts,
                                                  int64_t total, const FirTaps taps, float* __restrict__ out) {
  __shared__ int u_first;
  const int64_t base = (int64_t)blockIdx.x * 1024;
  if (threadIdx.x == 0) {
    int lo = 0, hi = n_utts - 1;  // last u with sample_off[u] <= base
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (sample_off[mid] <= base) lo = mid; else hi = mid - 1;
    }
    u_first = lo;
  }
  __syncthreads();
  int u = u_first;
  for (int j = 0; j < 4; ++j) {
    const int64_t n = base + threadIdx.x + 256 * j;
    if (n >= total) return;
    while (n >= sample_off[u + 1]) ++u;  // skips empty utterances too
    const int64_t s0 = sample_off[u], s1 = sample_off[u + 1];
    float acc = 0.f;
    for (int k = 0; k < taps.n; ++k) {
      const int64_t m = n + taps.delay - k;
      if (m >= s0 && m < s1) acc = fmaf(taps.w[k], (float)x[m], acc);
    }
    out[n] = acc;
  }
}

// Utterance that owns output row `row`: the last u with frame_off[u] <= row (utterances without
// frames share their offset with the next one and are skipped by taking the last).
__device__ __forceinline__ int utt_of_row(const int64_t* __restrict__ frame_off, int n_utts, int64_t row) {
  int lo = 0, hi = n_utts - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (frame_off[mid] <= row) lo = mid; else hi = mid - 1;
  }
  return lo;
}

// Context stacking, MLP_5hidlayer_40.py:70-105: one workgroup per output row t of
// utterance u; out[t, f*C + c] = feats[clamp(t + f - h), c].  The write of N*C contiguous
// floats per row is the traffic that matters (N x the read, which stays in L2): the row is
// written with 16-byte non-temporal stores (a few scalar stores bring an odd row width to
// 16-byte alignment first), the source values are gathered with 4-byte loads.
__global__ void __launch_bounds__(256) stack_context_kernel(const float* __restrict__ feats, int64_t feats_ld, int ncol,
                                                            int framenum, const int64_t* __restrict__ frame_off,
                                                            int n_utts, float* __restrict__ out) {
  const int64_t row = blockIdx.x;
  const int u = utt_of_row(frame_off, n_utts, row);  // every thread walks the same 17 or so steps (L1/L2-resident)
  const int64_t f0 = frame_off[u], f1 = frame_off[u + 1];
  const int half = (framenum - 1) / 2;
  const int width = framenum * ncol;
  float* const o = out + row * (int64_t)width;
  auto src_row = [&](int f) {
    int64_t t = row + f - half;
    t = t < f0 ? f0 : (t > f1 - 1 ? f1 - 1 : t);
    return feats + t * feats_ld;
  };
  const int lead = min(width, (int)((4 - (((uintptr_t)o >> 2) & 3)) & 3));  // floats up to the next 16-byte boundary
  if ((int)threadIdx.x < lead) {
    const int e = threadIdx.x, ff = e / ncol;
    o[e] = src_row(ff)[e - ff * ncol];
  }
  const int nvec = (width - lead) >> 2;
  // element index of this thread's chunk and its (frame, column) split, advanced incrementally
  int i = lead + 4 * (int)threadIdx.x;
  int f = i / ncol, c = i - f * ncol;
  const int di = 4 * (int)blockDim.x, df = di / ncol, dc = di - df * ncol;
  for (int v = threadIdx.x; v < nvec; v += blockDim.x) {
    float val[4];
    if (c + 3 < ncol) {
      const float* sp = src_row(f) + c;
      val[0] = sp[0]; val[1] = sp[1]; val[2] = sp[2]; val[3] = sp[3];
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        int cc = c + j, ff = f;
        while (cc >= ncol) { cc -= ncol; ++ff; }
        val[j] = src_row(ff)[cc];
      }
    }
    typedef float f32x4_t __attribute__((ext_vector_type(4)));
    f32x4_t pk = {val[0], val[1], val[2], val[3]};
    __builtin_nontemporal_store(pk, (f32x4_t*)(o + i));
    i += di;
    f += df;
    c += dc;
    if (c >= ncol) { c -= ncol; ++f; }
  }
  const int done = lead + 4 * nvec;
  if ((int)threadIdx.x < width - done) {
    const int e = done + threadIdx.x;
    const int ff = e / ncol;
    o[e] = src_row(ff)[e - ff * ncol];
  }
}

// The same operation for a 16-byte aligned output: a workgroup owns kStackRows consecutive output
// rows, i.e. one contiguous, 16-byte aligned run of kStackRows*N*C floats (8 rows of any width
// are a multiple of 4 floats), and streams it out with aligned non-temporal 16-byte stores, so
// there are few, long workgroups and no per-row alignment fix-ups.  Utterance bounds of the rows
// sit in LDS; (row, frame, column) of each thread's position advance incrementally.
constexpr int kStackRows = 8;
__global__ void __launch_bounds__(256) stack_context_rows_kernel(const float* __restrict__ feats, int64_t feats_ld, int ncol,
                                                                 int framenum, const int64_t* __restrict__ frame_off,
                                                                 int n_utts, float* __restrict__ out,
                                                                 int64_t total_rows) {
  __shared__ int64_t lo[kStackRows], hi[kStackRows];
  // Workgroups are handed to the 8 XCDs round-robin and each XCD has its own L2: give every XCD
  // one contiguous eighth of the rows, so that the N-fold re-read of a source row is served by
  // the L2 that fetched it instead of by the fabric eight times over.
  const int64_t nblk = (total_rows + kStackRows - 1) / kStackRows;
  const int64_t per_xcd = (nblk + 7) / 8;
  const int64_t blk = (int64_t)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  if (blk >= nblk) return;
  const int64_t row0 = blk * kStackRows;
  const int nrows = (int)(total_rows - row0 < kStackRows ? total_rows - row0 : kStackRows);
  if ((int)threadIdx.x < nrows) {
    const int u = utt_of_row(frame_off, n_utts, row0 + threadIdx.x);
    lo[threadIdx.x] = frame_off[u];
    hi[threadIdx.x] = frame_off[u + 1] - 1;
  }
  __syncthreads();
  const int half = (framenum - 1) / 2;
  const int width = framenum * ncol;
  const int total = nrows * width;
  const int nvec = total >> 2;
  float* const o = out + row0 * (int64_t)width;
  const float rcp = 1.0f / (float)ncol;
  auto src = [&](int r, int f) {
    int64_t t = row0 + r + f - half;
    t = t < lo[r] ? lo[r] : (t > hi[r] ? hi[r] : t);
    return feats + t * feats_ld;
  };
  // position of this thread's first element: row r, offset rem inside the row
  int i = 4 * (int)threadIdx.x;
  int r = i / width, rem = i - r * width;
  const int di = 4 * 256, dr = di / width, drem = di - dr * width;
  typedef float f32x4_t __attribute__((ext_vector_type(4)));
  for (int v = threadIdx.x; v < nvec; v += 256) {
    int f = (int)((float)rem * rcp);  // rem / ncol, fixed up below (rem < 2^24)
    if (f * ncol > rem) --f;
    if ((f + 1) * ncol <= rem) ++f;
    const int c = rem - f * ncol;
    float val[4];
    if (c + 3 < ncol) {
      // one 16-byte load at 4-byte alignment (global memory accepts it; an odd row width puts
      // the source anywhere)
      typedef float f32x4_u __attribute__((ext_vector_type(4), aligned(4)));
      const f32x4_u q4 = *(const f32x4_u*)(src(r, f) + c);
      val[0] = q4.x; val[1] = q4.y; val[2] = q4.z; val[3] = q4.w;
    } else {
      int rr = r, ff = f, cc = c;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        val[j] = src(rr, ff)[cc];
        if (++cc == ncol) {
          cc = 0;
          if (++ff == framenum) { ff = 0; ++rr; }
        }
        if (rr >= nrows) rr = nrows - 1;  // past the last element of the run: value unused
      }
    }
    f32x4_t pk = {val[0], val[1], val[2], val[3]};
    __builtin_nontemporal_store(pk, (f32x4_t*)(o + 4 * (int64_t)v));  // plain stores measured 40 % slower here
    r += dr;
    rem += drem;
    if (rem >= width) { rem -= width; ++r; }
  }
  if ((int)threadIdx.x < total - 4 * nvec) {  // fewer than 4 floats at the very end of the array
    const int e = 4 * nvec + threadIdx.x;
    const int rr = e / width, re = e - rr * width, ff = re / ncol;
    o[e] = src(rr, ff)[re - ff * ncol];
  }
}

// Context stacking in its general form: output row i holds the window of stacked row
// t = row_index ? row_index[i] : i, in one of two element orders:
//   frame-major  out[i, f*C + c]   (the MLP vector, MLP_5hidlayer_40.py:101-102)
//   tile         out[i, c*N + f]   (the [C, N] slice paddeddata[:, pos:pos+framenum] the CNN
//                                   scripts keep, CNN_DenseNet40_6.py:98; row_index carries their
//                                   shuffled window order, :91-97)
// One workgroup per output row: the N source rows go through LDS so that both the reads and
// the (transposed) writes are contiguous.
__global__ void __launch_bounds__(256) stack_context_gather_kernel(const float* __restrict__ feats, int64_t feats_ld, int ncol,
                                                                   int framenum, const int64_t* __restrict__ frame_off,
                                                                   int n_utts, int64_t total_rows,
                                                                   const int64_t* __restrict__ row_index, int tile_layout,
                                                                   float* __restrict__ out) {
  extern __shared__ float win[];  // [N][pitch]
  const int pitch = ncol | 1;     // odd pitch: the transposed read walks all banks
  int64_t row = row_index ? row_index[blockIdx.x] : (int64_t)blockIdx.x;
  row = row < 0 ? 0 : (row > total_rows - 1 ? total_rows - 1 : row);  // a bad index must not fault
  const int u = utt_of_row(frame_off, n_utts, row);
  const int64_t f0 = frame_off[u], f1 = frame_off[u + 1];
  const int half = (framenum - 1) / 2;
  const int width = framenum * ncol;
  for (int e = threadIdx.x; e < width; e += 256) {
    const int f = e / ncol, c = e - f * ncol;
    int64_t t = row + f - half;
    t = t < f0 ? f0 : (t > f1 - 1 ? f1 - 1 : t);
    win[f * pitch + c] = feats[t * feats_ld + c];
  }
  __syncthreads();
  float* const o = out + (int64_t)blockIdx.x * width;
  if (tile_layout) {
    for (int e = threadIdx.x; e < width; e += 256) {
      const int c = e / framenum, f = e - c * framenum;
      o[e] = win[f * pitch + c];
    }
  } else {
    for (int e = threadIdx.x; e < width; e += 256) {
      const int f = e / ncol, c = e - f * ncol;
      o[e] = win[f * pitch + c];
    }
  }
}

// CMVN statistics, float64, bit-reproducible: no cross-workgroup atomics anywhere.
//   stage 1  workgroup b reduces the rows [b*rows_per_block, (b+1)*rows_per_block) of x and writes
//            its column sums and sums of squares to partial[b][0 .. 2*ncol);
//   stage 2  workgroup c adds partial[0 .. nblk)[c] in an order fixed by the code (a strided
//            sweep per thread, then a fixed pairwise tree) and adds the total into sums[1 + c];
//            the row count goes to sums[0].
// Every launch therefore produces the same bits for the same input, whatever the timing.
constexpr int kCmvnMaxBlocks = 1024;  // stage-1 workgroups at most (bounds the partials scratch)

__global__ void cmvn_partial_kernel(const float* __restrict__ x, int64_t ld, int64_t n_rows, int ncol,
                                    int64_t rows_per_block, double* __restrict__ partial) {
  extern __shared__ double red[];  // [2][blockDim.x]
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = r0 + rows_per_block < n_rows ? r0 + rows_per_block : n_rows;
  // thread (ty, c): c = column handled, ty = row phase
  const int nphase = blockDim.x / ncol;
  const int c = threadIdx.x % ncol, ty = threadIdx.x / ncol;
  double s = 0.0, ss = 0.0;
  if (ty < nphase) {
    for (int64_t r = r0 + ty; r < r1; r += nphase) {
      const double v = (double)x[r * ld + c];
      s += v;
      ss += v * v;
    }
  }
  red[threadIdx.x] = s;
  red[blockDim.x + threadIdx.x] = ss;
  __syncthreads();
  if ((int)threadIdx.x < ncol) {
    double a = 0.0, b = 0.0;
    for (int y = 0; y < nphase; ++y) {  // fixed order
      a += red[y * ncol + threadIdx.x];
      b += red[blockDim.x + y * ncol + threadIdx.x];
    }
    double* const p = partial + (int64_t)blockIdx.x * 2 * ncol;
    p[threadIdx.x] = a;
    p[ncol + threadIdx.x] = b;
  }
}

// Stage 1 for rows that lie back to back (ld == ncol, 16-byte aligned base): the array is one
// stream of 16-byte pieces, thread g of the G = 256 * gridDim.x threads takes the pieces g, g + G,
// g + 2G, ...  The host picks G so that 4 G is a multiple of ncol: the four elements of every piece
// of a thread then belong to the same four columns ((4 g + k) mod ncol), and the thread keeps one
// float64 sum and sum of squares for each.  Loads are whole 16-byte pieces, coalesced, several in
// flight.  The workgroup's 1024 sums meet in LDS, where element e belongs to column
// (1024 blockIdx.x + e) mod ncol; column c is added up in ascending e.  Same fixed order on every
// launch, no atomics.
__global__ void __launch_bounds__(256) cmvn_partial_flat_kernel(const float* __restrict__ x, int64_t n_elems, int ncol,
                                                                double* __restrict__ partial) {
  __shared__ double red[2][1024];
  const int64_t G = (int64_t)gridDim.x * 256, g = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t n4 = n_elems >> 2;
  const float4* const x4 = (const float4*)__builtin_assume_aligned(x, 16);
  double s[4] = {0.0, 0.0, 0.0, 0.0}, ss[4] = {0.0, 0.0, 0.0, 0.0};
  auto add = [&](const float4 v) {
    const double a = v.x, b = v.y, c = v.z, d = v.w;
    s[0] += a; ss[0] = fma(a, a, ss[0]);
    s[1] += b; ss[1] = fma(b, b, ss[1]);
    s[2] += c; ss[2] = fma(c, c, ss[2]);
    s[3] += d; ss[3] = fma(d, d, ss[3]);
  };
  int64_t i = g;
  for (; i + 7 * G < n4; i += 8 * G) {  // eight pieces in flight
    float4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = x4[i + u * G];
#pragma unroll
    for (int u = 0; u < 8; ++u) add(v[u]);
  }
  for (; i < n4; i += G) add(x4[i]);
  if (i == n4) {  // the last, incomplete piece (n_elems % 4 elements) falls to the thread next in turn
    for (int k = 0; k < (int)(n_elems & 3); ++k) {
      const double a = x[4 * n4 + k];
      s[k] += a;
      ss[k] = fma(a, a, ss[k]);
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    red[0][4 * threadIdx.x + k] = s[k];
    red[1][4 * threadIdx.x + k] = ss[k];
  }
  __syncthreads();
  const int off = (int)(((int64_t)blockIdx.x * 1024) % ncol);
  for (int c = threadIdx.x; c < ncol; c += 256) {
    double a = 0.0, b = 0.0;
    for (int e = c >= off ? c - off : c - off + ncol; e < 1024; e += ncol) {
      a += red[0][e];
      b += red[1][e];
    }
    double* const p = partial + (int64_t)blockIdx.x * 2 * ncol;
    p[c] = a;
    p[ncol + c] = b;
  }
}

__global__ void __launch_bounds__(256) cmvn_finish_kernel(const double* __restrict__ partial, int nblk, int ncol,
                                                          int64_t n_rows, double* __restrict__ sums) {
  __shared__ double red[256];
  const int c = blockIdx.x;  // 0 .. 2*ncol-1
  double a = 0.0;
  for (int b = threadIdx.x; b < nblk; b += 256) a += partial[(int64_t)b * 2 * ncol + c];
  red[threadIdx.x] = a;
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) {
    if ((int)threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    sums[1 + c] += red[0];
    if (c == 0) sums[0] += (double)n_rows;
  }
}

// Second stage for the column sums the fused kernel leaves per tile (kOutMfcc39Sums): tile_sums is
// [n_tiles][7][16] float -- blocks of 16 coefficients: sums of c - K, delta-delta, delta, then of the squares of
// c - K, delta, delta-delta, then the tile's pivots K.  Workgroup (k, q), k = 0..2 the sums of c | delta |
// delta-delta and 3..5 their squares, adds its column over the tiles in float64, in an order fixed by the code,
// and puts the pivot back: sum x = S1 + n K, sum x^2 = S2 + 2 K S1 + n K^2.
__global__ void __launch_bounds__(256) tile_sums_finish_kernel(const float* __restrict__ tile_sums, const Tile* __restrict__ tiles,
                                                               int64_t n_tiles, int ncep, int64_t n_rows, double* __restrict__ sums) {
  __shared__ double red[256];
  const int k = blockIdx.x / ncep, q = blockIdx.x - k * ncep;  // k: 0..2 sums, 3..5 sums of squares
  double a = 0.0;
  for (int64_t b = threadIdx.x; b < n_tiles; b += 256) {
    const float* const ts = tile_sums + b * kTileSums;
    const int block = k == 1 ? 2 : (k == 2 ? 1 : k);  // the kernel leaves delta-delta before delta
    double v = (double)ts[16 * block + q];
    if (k == 0 || k == 3) {
      const double piv = (double)ts[kTileSumCols + q], n = (double)tiles[b].nt;
      v = k == 0 ? v + n * piv : v + 2.0 * piv * (double)ts[q] + n * piv * piv;
    }
    a += v;
  }
  red[threadIdx.x] = a;
  __syncthreads();
  for (int m = 128; m > 0; m >>= 1) {
    if ((int)threadIdx.x < m) red[threadIdx.x] += red[threadIdx.x + m];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    sums[1 + k * ncep + q] += red[0];  // the 3 ncep sums, then the 3 ncep sums of squares
    if (blockIdx.x == 0) sums[0] += (double)n_rows;
  }
}

// (x + mm) .* vv in place (the form of anal_recog_nn_liste.m:180).  A workgroup owns a slab of
// rows; consecutive threads touch consecutive elements and track (row, column) incrementally,
// so there is no division per element.  mm / vv sit in LDS.
__global__ void __launch_bounds__(256) cmvn_apply_kernel(float* __restrict__ x, int64_t ld, int64_t n_rows, int ncol,
                                                         const float* __restrict__ mm, const float* __restrict__ vv,
                                                         int rows_per_block) {
  extern __shared__ float cm[];  // [2][ncol]
  for (int i = threadIdx.x; i < ncol; i += blockDim.x) {
    cm[i] = mm[i];
    cm[ncol + i] = vv[i];
  }
  __syncthreads();
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int nr = (int)(n_rows - r0 < rows_per_block ? n_rows - r0 : rows_per_block);
  const int total = nr * ncol;
  int r = threadIdx.x / ncol, c = threadIdx.x - r * ncol;
  const int dr = blockDim.x / ncol, dc = blockDim.x - dr * ncol;
  float* const base = x + r0 * ld;
  for (int i = threadIdx.x; i < total; i += blockDim.x) {
    float* p = base + (int64_t)r * ld + c;
    *p = (*p + cm[c]) * cm[ncol + c];
    r += dr;
    c += dc;
    if (c >= ncol) { c -= ncol; ++r; }
  }
}

// The same update for rows that lie back to back (ld == ncol, 16-byte aligned base), as a stream
// of 16-byte pieces: with 4 * (number of threads) a multiple of ncol every piece of a thread has
// the same four columns, so their constants stay in registers and a piece is one load, four
// multiply-adds' worth of work and one store.
__global__ void __launch_bounds__(256) cmvn_apply_flat_kernel(float* __restrict__ x, int64_t n_elems, int ncol,
                                                              const float* __restrict__ mm, const float* __restrict__ vv) {
  const int64_t G = (int64_t)gridDim.x * 256, g = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t n4 = n_elems >> 2;
  float m[4], v[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int c = (int)((4 * g + k) % ncol);
    m[k] = mm[c];
    v[k] = vv[c];
  }
  float4* const x4 = (float4*)__builtin_assume_aligned(x, 16);
  auto upd = [&](float4 a) { return make_float4((a.x + m[0]) * v[0], (a.y + m[1]) * v[1], (a.z + m[2]) * v[2], (a.w + m[3]) * v[3]); };
  int64_t i = g;
  for (; i + 3 * G < n4; i += 4 * G) {
    const float4 a0 = x4[i], a1 = x4[i + G], a2 = x4[i + 2 * G], a3 = x4[i + 3 * G];
    x4[i] = upd(a0); x4[i + G] = upd(a1); x4[i + 2 * G] = upd(a2); x4[i + 3 * G] = upd(a3);
  }
  for (; i < n4; i += G) x4[i] = upd(x4[i]);
  if (i == n4)
    for (int k = 0; k < (int)(n_elems & 3); ++k) x[4 * n4 + k] = (x[4 * n4 + k] + m[k]) * v[k];
}

// Per-utterance scalar normalisation (x - mean) / std with mean and population std taken over
// ALL values of the utterance, every frame and every coefficient (inference_ctc_eri_05.py:291,
// CTC/ctc_test_1.py:54).  One workgroup per utterance: float64 sums in LDS, then the in-place
// update (the second read comes from L2).
__global__ void __launch_bounds__(256) utt_scalar_norm_kernel(float* __restrict__ x, int64_t ld, int ncol,
                                                              const int64_t* __restrict__ frame_off) {
  __shared__ double red[2][256];
  const int u = blockIdx.x;
  const int64_t r0 = frame_off[u], nr = frame_off[u + 1] - r0;
  const int64_t total = nr * ncol;
  float* const base = x + r0 * ld;
  double s = 0.0, ss = 0.0;
  {
    int64_t r = threadIdx.x / ncol;
    int c = threadIdx.x - (int)r * ncol;
    const int dr = blockDim.x / ncol, dc = blockDim.x - dr * ncol;
    for (int64_t i = threadIdx.x; i < total; i += blockDim.x) {
      const double v = base[r * ld + c];
      s += v;
      ss += v * v;
      r += dr;
      c += dc;
      if (c >= ncol) { c -= ncol; ++r; }
    }
  }
  red[0][threadIdx.x] = s;
  red[1][threadIdx.x] = ss;
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) {
    if ((int)threadIdx.x < k) {
      red[0][threadIdx.x] += red[0][threadIdx.x + k];
      red[1][threadIdx.x] += red[1][threadIdx.x + k];
    }
    __syncthreads();
  }
  if (total == 0) return;
  const double mean = red[0][0] / (double)total;
  const double var = red[1][0] / (double)total - mean * mean;
  const float m = (float)mean, inv = (float)(1.0 / sqrt(var > 0.0 ? var : 0.0));
  {
    int64_t r = threadIdx.x / ncol;
    int c = threadIdx.x - (int)r * ncol;
    const int dr = blockDim.x / ncol, dc = blockDim.x - dr * ncol;
    for (int64_t i = threadIdx.x; i < total; i += blockDim.x) {
      float* p = base + r * ld + c;
      *p = (*p - m) * inv;
      r += dr;
      c += dc;
      if (c >= ncol) { c -= ncol; ++r; }
    }
  }
}

// Ragged rows -> dense [B, T, ncol] batch, padded with `value` and truncated to T frames
// (pad_sequences, CTC/utils.py:71-128; the zero-padded [batch, time_steps, coeffs] arrays of
// BiLSTM_7.py:82-124).  pre_pad / pre_trunc select 'pre' instead of 'post'.
__global__ void __launch_bounds__(256) pad_batch_kernel(const float* __restrict__ feats, int64_t ld, int ncol,
                                                        const int64_t* __restrict__ frame_off, int T, float value,
                                                        int pre_pad, int pre_trunc, float* __restrict__ out) {
  const int u = blockIdx.y;
  const int64_t r0 = frame_off[u];
  const int nr = (int)(frame_off[u + 1] - r0);
  const int kept = nr < T ? nr : T;
  const int src0 = pre_trunc ? nr - kept : 0;  // first source frame kept
  const int dst0 = pre_pad ? T - kept : 0;     // where it lands
  const int total = T * ncol;
  float* const o = out + (int64_t)u * total;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    const int t = i / ncol, c = i - t * ncol;
    const int k = t - dst0;
    o[i] = (k >= 0 && k < kept) ? feats[(r0 + src0 + k) * ld + c] : value;
  }
}

}  // namespace mfcc
