// One kernel per reference stage, for callers that drive the pipeline function by
// function (the reference's main(), mfcc-self5.py:211-229) and for stage-level parity
// tests.  These are simple bandwidth-shaped kernels; the fused kernel is the hot path.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

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

// Context stacking, MLP_5hidlayer_40.py:70-105: one workgroup per output row t of
// utterance u; out[t, f*C + c] = feats[clamp(t + f - h), c].  The write of N*C contiguous
// floats per row is the traffic that matters (17 x the read).
__global__ void stack_context_kernel(const float* __restrict__ feats, int64_t feats_ld, int ncol, int framenum,
                                     const int64_t* __restrict__ frame_off, const int32_t* __restrict__ row_utt,
                                     float* __restrict__ out) {
  const int64_t row = blockIdx.x;
  const int u = row_utt[row];
  const int64_t f0 = frame_off[u], f1 = frame_off[u + 1];
  const int half = (framenum - 1) / 2;
  const int width = framenum * ncol;
  float* o = out + row * (int64_t)width;
  for (int i = threadIdx.x; i < width; i += blockDim.x) {
    const int f = i / ncol, c = i - f * ncol;
    int64_t t = row + f - half;
    t = t < f0 ? f0 : (t > f1 - 1 ? f1 - 1 : t);
    o[i] = feats[t * feats_ld + c];
  }
}

// sums[0] += rows, sums[1+c] += sum x, sums[1+D+c] += sum x^2, all in float64.
// Each workgroup reduces a slab of rows in registers/LDS and issues one atomic per column.
__global__ void cmvn_accumulate_kernel(const float* __restrict__ x, int64_t ld, int64_t n_rows, int ncol,
                                       int64_t rows_per_block, double* __restrict__ sums) {
  extern __shared__ double red[];  // [2][blockDim.x]
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = r0 + rows_per_block < n_rows ? r0 + rows_per_block : n_rows;
  // thread (ty, c): c = column handled, ty = row phase
  const int cols_pad = ncol;
  const int nphase = blockDim.x / cols_pad;
  const int c = threadIdx.x % cols_pad, ty = threadIdx.x / cols_pad;
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
  if (threadIdx.x < ncol) {
    double a = 0.0, b = 0.0;
    for (int y = 0; y < nphase; ++y) {  // fixed order: deterministic per block
      a += red[y * cols_pad + threadIdx.x];
      b += red[blockDim.x + y * cols_pad + threadIdx.x];
    }
    atomicAdd(&sums[1 + threadIdx.x], a);
    atomicAdd(&sums[1 + ncol + threadIdx.x], b);
  }
  if (threadIdx.x == 0) atomicAdd(&sums[0], (double)(r1 - r0));
}

__global__ void cmvn_apply_kernel(float* __restrict__ x, int64_t ld, int64_t n_rows, int ncol,
                                  const float* __restrict__ mm, const float* __restrict__ vv) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n_rows * ncol) return;
  const int64_t r = idx / ncol;
  const int c = (int)(idx - r * ncol);
  float* p = x + r * ld + c;
  *p = (*p + mm[c]) * vv[c];  // (x + mm) .* vv, anal_recog_nn_liste.m:180
}

}  // namespace mfcc
