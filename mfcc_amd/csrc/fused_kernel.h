// Fused MFCC kernel for gfx950: int16/float audio in HBM -> feature rows in HBM,
// one launch, every intermediate in LDS or registers.
//
// Work decomposition
//   grid      one workgroup (256 threads = 4 waves) per TILE of <= kTileFrames output
//             frames of one utterance; tiles never cross utterance boundaries because
//             y[0]=x[0] (mfcc-self5.py:68), the zero tail (:89-90) and the delta edge
//             replication (:174) are all per utterance.
//   batch     a tile is walked in batches of 16 frames.  Per batch:
//               1. all threads stage 15*160+400 samples (converted to float, pre-emphasised)
//                  into LDS -- each sample is read from HBM once per batch although it
//                  belongs to 2.5 frames;
//               2. each wave transforms 4 frames, 16 lanes per frame (fft_lane.h), and leaves
//                  the 257 spectrum values of each frame in LDS;
//               3. mel filterbank + ln, then DCT*lifter; statics accumulate in an LDS tile.
//   epilogue  delta and delta-delta over the LDS tile (4 halo frames were computed on each
//             side where the utterance continues), rows written to HBM.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fft_lane.h"

namespace mfcc {

constexpr int kFrameSize = 400;   // int(16000 * 0.025), mfcc-self5.py:75
constexpr int kFrameStep = 160;   // int(16000 * 0.010), mfcc-self5.py:76
constexpr int kNfft = 512;        // next power of two, mfcc-self5.py:79-84
constexpr int kNbin = kNfft / 2 + 1;
constexpr int kBatch = 16;        // frames per batch (4 waves x 4 frames)
constexpr int kTileFrames = 152;  // max output frames per workgroup
constexpr int kHalo = 4;          // delta-delta reach: 2 * pad_fac
constexpr int kStaticMax = kTileFrames + 2 * kHalo;  // 160 = 10 batches
constexpr int kMaxCep = 16;
constexpr int kMaxMel = 47;
constexpr int kMelLd = 260;       // row stride of the mel weight table
constexpr int kStage = (kBatch - 1) * kFrameStep + kFrameSize;  // 2800 samples per batch
constexpr int kLogLd = 48;        // row stride of the per-batch ln-mel rows (nmel + 1 <= 48)

// LDS carve (floats)
constexpr int kXbufFloats = kBatch * 512;            // exchange buffers, later spectrum rows, later deltas
constexpr int kAbufFloats = 2816;                    // staged audio, later ln-mel rows
constexpr int kEbufFloats = 16;                      // per-frame energy sums
constexpr int kSbufFloats = kStaticMax * kMaxCep;    // static features of the tile (row stride = numcep)
constexpr int kSmemFloats = kXbufFloats + kAbufFloats + kEbufFloats + kSbufFloats;
constexpr int kSpecRowStride = 514;                  // spectrum row of frame f starts at f*514 floats

constexpr float kEps64 = 2.220446049250313e-16f;     // np.finfo(float).eps, mfcc-self5.py:131

struct Tile {
  int32_t utt;  // utterance index
  int32_t t0;   // first output frame (utterance-relative)
  int32_t nt;   // number of output frames
  int32_t pad;
};

struct FusedParams {
  const void* audio;
  const int64_t* sample_off;  // [n_utts + 1]
  const int64_t* frame_off;   // [n_utts + 1]
  const Tile* tiles;
  float* out;
  int64_t out_ld;
  const float* tw512;     // [512][2]  exp(-2 pi i m / 512)
  const float* melw;      // [nmel][kMelLd] filter weights times spec_scale
  const int2* melrange;   // [nmel] {first non-zero bin, count}
  const float* dctw;      // [nmel][kMaxCep] DCT-II ortho times lifter
  const float* window;    // [512] or nullptr
  int32_t nmel;
  int32_t numcep;
  int32_t audio_dtype;    // 0 = int16, 1 = float32
  int32_t use_preemph;
  float pre_hi, pre_lo;   // coefficient split in two floats (hi + lo ~ double value)
  int32_t spectrum;       // 0 power, 1 magnitude
  int32_t floor_mel;
  float spec_scale;       // 1/(4*nfft) for power, 1/2 for magnitude (both exact powers of two)
};

enum : int { kOutMfcc39 = 0, kOutStatic = 1, kOutLogmel = 2, kOutSpec = 3, kOutLogspec = 4 };

__device__ __forceinline__ void wave_sync() {
  // LDS operations of one wave execute in program order; this only stops the compiler
  // from moving LDS accesses across the exchange point.
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ float lane_read(int byte_addr, float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(byte_addr, __builtin_bit_cast(int, v)));
}

template <int OUT>
__global__ void __launch_bounds__(256, 3) fused_kernel(const FusedParams p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* const xbuf = smem;
  float* const abuf = xbuf + kXbufFloats;
  float* const ebuf = abuf + kAbufFloats;
  float* const sbuf = ebuf + kEbufFloats;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int q = lane & 15;
  const int fb = wave * 4 + (lane >> 4);  // frame slot inside the batch

  const Tile tile = p.tiles[blockIdx.x];
  const int64_t s_off = p.sample_off[tile.utt];
  const int64_t len = p.sample_off[tile.utt + 1] - s_off;
  const int64_t f_off = p.frame_off[tile.utt];
  const int nframes = (int)(p.frame_off[tile.utt + 1] - f_off);
  const int t0 = tile.t0, nt = tile.nt;
  const int s_lo = (OUT == kOutMfcc39) ? max(0, t0 - kHalo) : t0;
  const int s_hi = (OUT == kOutMfcc39) ? min(nframes, t0 + nt + kHalo) : t0 + nt;
  const int nmel = p.nmel, ncep = p.numcep;

  // per-lane twiddles: stage 1 uses W256^(q k1) = W512^(2 q k1), the untangle W512^(q + 16 s)
  float tw1r[16], tw1i[16], tw2r[8], tw2i[8];
#pragma unroll
  for (int k1 = 1; k1 < 16; ++k1) {
    const int m = (2 * q * k1) & 511;
    tw1r[k1] = p.tw512[2 * m];
    tw1i[k1] = p.tw512[2 * m + 1];
  }
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    const int m = q + 16 * s;
    tw2r[s] = p.tw512[2 * m];
    tw2i[s] = p.tw512[2 * m + 1];
  }
  const int partner_addr = ((lane & 48) | partner_lane(q)) << 2;
  const bool mag = p.spectrum != 0;

  for (int tb = s_lo; tb < s_hi; tb += kBatch) {
    // ---- 1. stage audio: float conversion, pre-emphasis, zero tail -------------------
    {
      const int64_t base = (int64_t)tb * kFrameStep;
      for (int i = tid; i < kStage; i += 256) {
        const int64_t s = base + i;
        float v = 0.f;
        if (s < len) {
          if (p.audio_dtype == 0) {
            const int16_t* a = (const int16_t*)p.audio + s_off;
            const float x0 = (float)a[s];
            if (p.use_preemph && s > 0) {
              const float x1 = (float)a[s - 1];
              v = fmaf(-p.pre_lo, x1, fmaf(-p.pre_hi, x1, x0));
            } else {
              v = x0;
            }
          } else {
            // float input: numpy evaluates x[n] - coeff*x[n-1] in float32 with two roundings
            const float* a = (const float*)p.audio + s_off;
            const float x0 = a[s];
            v = (p.use_preemph && s > 0) ? __fsub_rn(x0, __fmul_rn(p.pre_hi, a[s - 1])) : x0;
          }
        }
        abuf[i] = v;
      }
    }
    __syncthreads();

    // ---- 2. FFT-512 of 4 frames per wave, 16 lanes per frame ------------------------
    {
      const float* fa = abuf + fb * kFrameStep;
      float xr[16], xi[16];
#pragma unroll
      for (int n1 = 0; n1 < 16; ++n1) {
        const int n = 16 * n1 + q;  // complex sample index: (x[2n], x[2n+1])
        if (2 * (16 * n1 + 15) + 1 < kFrameSize) {
          const float2 v = *(const float2*)(fa + 2 * n);
          xr[n1] = v.x; xi[n1] = v.y;
        } else if (2 * (16 * n1) < kFrameSize) {
          const float2 v = *(const float2*)(fa + 2 * min(n, kFrameSize / 2 - 1));
          const bool ok = 2 * n + 1 < kFrameSize;
          xr[n1] = ok ? v.x : 0.f; xi[n1] = ok ? v.y : 0.f;
        } else {
          xr[n1] = 0.f; xi[n1] = 0.f;
        }
      }
      if (p.window) {
#pragma unroll
        for (int n1 = 0; n1 < 16; ++n1) {
          if (2 * (16 * n1) < kFrameSize) {
            const float2 w = *(const float2*)(p.window + 2 * (16 * n1 + q));
            xr[n1] *= w.x; xi[n1] *= w.y;
          }
        }
      }
      fft16(xr, xi);
      float2* const T = (float2*)(xbuf + fb * 512);
#pragma unroll
      for (int k1 = 0; k1 < 16; ++k1) {
        if (k1) cmul(xr[k1], xi[k1], tw1r[k1], tw1i[k1]);
        T[xpose_index(k1, q)] = make_float2(xr[k1], xi[k1]);
      }
      wave_sync();
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float4 v = *(const float4*)(T + xpose_index(q, 2 * j));
        xr[2 * j] = v.x; xi[2 * j] = v.y; xr[2 * j + 1] = v.z; xi[2 * j + 1] = v.w;
      }
      fft16(xr, xi);
      wave_sync();  // every lane of this wave has read its row before spectrum rows overwrite it

      float* const P = xbuf + fb * kSpecRowStride;
      float esum = 0.f;
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        float br = lane_read(partner_addr, xr[15 - s]);
        float bi = lane_read(partner_addr, xi[15 - s]);
        if (q == 0) { br = xr[(16 - s) & 15]; bi = xi[(16 - s) & 15]; }
        float lo, hi;
        untangle_power(xr[s], xi[s], br, bi, tw2r[s], tw2i[s], lo, hi);
        if (mag) { lo = __builtin_sqrtf(lo); hi = __builtin_sqrtf(hi); }
        P[q + 16 * s] = lo;
        P[256 - q - 16 * s] = hi;
        esum += lo + hi;
      }
      if (q == 0) {  // bin 128 pairs with itself: X[128] = conj Z[128]
        float m = 4.f * (xr[8] * xr[8] + xi[8] * xi[8]);
        if (mag) m = __builtin_sqrtf(m);
        P[128] = m;
        esum += m;
      }
      esum += __shfl_xor(esum, 1, 16);
      esum += __shfl_xor(esum, 2, 16);
      esum += __shfl_xor(esum, 4, 16);
      esum += __shfl_xor(esum, 8, 16);
      if (q == 0) ebuf[fb] = esum;
    }
    __syncthreads();

    if (OUT == kOutSpec || OUT == kOutLogspec) {
      const int nvalid = min(kBatch, s_hi - tb);
      for (int idx = tid; idx < nvalid * kNbin; idx += 256) {
        const int f = idx / kNbin, k = idx - f * kNbin;
        float v = xbuf[f * kSpecRowStride + k] * p.spec_scale;
        if (OUT == kOutLogspec) v = logf(v);
        p.out[(f_off + tb + f) * p.out_ld + k] = v;
      }
      // the next batch rewrites xbuf only after its staging barrier
      continue;
    }

    // ---- 3. mel filterbank + ln, ln energy -------------------------------------------
    {
      for (int item = tid; item < kBatch * nmel; item += 256) {
        const int f = item & 15, j = item >> 4;
        const int2 rg = p.melrange[j];
        const float* w = p.melw + j * kMelLd + rg.x;
        const float* P = xbuf + f * kSpecRowStride + rg.x;
        float acc = 0.f;
        for (int i = 0; i < rg.y; ++i) acc = fmaf(P[i], w[i], acc);
        if (p.floor_mel && acc == 0.f) acc = kEps64;
        abuf[f * kLogLd + j] = logf(acc);  // unfloored: ln 0 = -inf, as mfcc-self5.py:127
      }
      if (tid < kBatch) {
        float e = ebuf[tid] * p.spec_scale;
        if (e == 0.f) e = kEps64;  // mfcc-self5.py:131
        abuf[tid * kLogLd + nmel] = logf(e);
      }
    }
    __syncthreads();

    if (OUT == kOutLogmel) {
      const int nvalid = min(kBatch, s_hi - tb);
      const int w = nmel + 1;
      for (int idx = tid; idx < nvalid * w; idx += 256) {
        const int f = idx / w, j = idx - f * w;
        p.out[(f_off + tb + f) * p.out_ld + j] = abuf[f * kLogLd + j];
      }
      __syncthreads();  // abuf is restaged next
      continue;
    }

    // ---- 4. DCT-II * lifter, c0 := ln E  (mfcc-self5.py:136-167) ------------------------
    for (int item = tid; item < kBatch * ncep; item += 256) {
      const int f = item & 15, c = item >> 4;
      const int t = tb + f;
      if (t < s_hi) {
        float v;
        if (c == 0) {
          v = abuf[f * kLogLd + nmel];
        } else {
          v = 0.f;
          for (int j = 0; j < nmel; ++j) v = fmaf(abuf[f * kLogLd + j], p.dctw[j * kMaxCep + c], v);
        }
        sbuf[(t - s_lo) * ncep + c] = v;
      }
    }
    __syncthreads();
  }

  if (OUT == kOutStatic) {
    for (int idx = tid; idx < nt * ncep; idx += 256) {
      const int r = idx / ncep, c = idx - r * ncep;
      p.out[(f_off + t0 + r) * p.out_ld + c] = sbuf[idx];
    }
  }

  if (OUT == kOutMfcc39) {
    // ---- 5. delta over the tile (+2 halo rows), then rows out -------------------------
    float* const dbuf = xbuf;
    const int d_lo = max(0, t0 - 2), d_hi = min(nframes, t0 + nt + 2);
    const int last = nframes - 1;
    for (int idx = tid; idx < (d_hi - d_lo) * ncep; idx += 256) {
      const int r = idx / ncep, c = idx - r * ncep;
      const int t = d_lo + r;
      const float a = sbuf[(max(t - 2, 0) - s_lo) * ncep + c];
      const float b = sbuf[(max(t - 1, 0) - s_lo) * ncep + c];
      const float m = sbuf[(t - s_lo) * ncep + c];
      const float d = sbuf[(min(t + 1, last) - s_lo) * ncep + c];
      const float e = sbuf[(min(t + 2, last) - s_lo) * ncep + c];
      // (-2a - b + 0m + d + 2e) / 10; the zero-weight centre tap keeps NaN propagation
      // identical to np.dot over the 5-row window (mfcc-self5.py:177-178)
      dbuf[idx] = (2.f * (e - a) + (d - b) + 0.f * m) * 0.1f;
    }
    __syncthreads();
    const int width = 3 * ncep;
    for (int idx = tid; idx < nt * width; idx += 256) {
      const int r = idx / width, c = idx - r * width;
      const int t = t0 + r;
      float v;
      if (c < ncep) {
        v = sbuf[(t - s_lo) * ncep + c];
      } else if (c < 2 * ncep) {
        v = dbuf[(t - d_lo) * ncep + (c - ncep)];
      } else {
        const int cc = c - 2 * ncep;
        const float a = dbuf[(max(t - 2, 0) - d_lo) * ncep + cc];
        const float b = dbuf[(max(t - 1, 0) - d_lo) * ncep + cc];
        const float m = dbuf[(t - d_lo) * ncep + cc];
        const float d = dbuf[(min(t + 1, last) - d_lo) * ncep + cc];
        const float e = dbuf[(min(t + 2, last) - d_lo) * ncep + cc];
        v = (2.f * (e - a) + (d - b) + 0.f * m) * 0.1f;
      }
      p.out[(f_off + t) * p.out_ld + c] = v;
    }
  }
}

}  // namespace mfcc
