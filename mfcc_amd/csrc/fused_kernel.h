// Fused MFCC kernel for gfx950: int16/float audio in HBM -> feature rows in HBM,
// one launch, every intermediate in LDS or registers.
//
// Work decomposition
//   grid      one workgroup (256 threads = 4 waves) per TILE of consecutive output frames of
//             one utterance (up to 19 batches, ~300 frames); tiles never cross utterance
//             boundaries because y[0]=x[0] (mfcc-self5.py:68), the zero tail (:89-90) and the
//             delta edge replication (:174) are all per utterance.
//   batch     a tile is walked in batches of 16 frames, two workgroup barriers per batch:
//               1. all threads stage 15*160+400 samples (aligned 16-byte loads, converted to
//                  float, pre-emphasised) into LDS -- each sample is read from HBM once per
//                  batch although it belongs to 2.5 frames; batch b+1 is staged while the
//                  matrix cores work on batch b;
//               2. each wave transforms 4 frames, 16 lanes per frame (fft_lane.h), and leaves
//                  the 257 spectrum values of each frame in LDS;                [barrier]
//               3. the [16 x 257] x [257 x (nmel+1)] filterbank product runs on the matrix
//                  cores (v_mfma_f32_16x16x4_f32, an exact f32 fma chain): every wave owns a
//                  group of filter columns, so it takes ln of its sums at once;    [barrier]
//               4. one wave (rotating) runs the [16 x nmel] x [nmel x numcep] DCT*lifter
//                  product on the matrix cores, sets c0 := ln E and writes the static
//                  features straight to their final place in the output rows.
//   epilogue  the tile's static columns are read back (L2-hot) into LDS, delta and
//             delta-delta are formed (4 halo frames were computed on each side where the
//             utterance continues) and the remaining 2*numcep columns are written.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fft_lane.h"

#ifndef MFCC_MIN_WAVES
#define MFCC_MIN_WAVES 3  // waves per SIMD the register allocation must allow (3 workgroups per CU)
#endif

namespace mfcc {

constexpr int kFrameSize = 400;   // int(16000 * 0.025), mfcc-self5.py:75
constexpr int kFrameStep = 160;   // int(16000 * 0.010), mfcc-self5.py:76
constexpr int kNfft = 512;        // next power of two, mfcc-self5.py:79-84
constexpr int kNbin = kNfft / 2 + 1;
constexpr int kBatch = 16;        // frames per batch (4 waves x 4 frames)
constexpr int kHalo = 4;          // delta-delta reach: 2 * pad_fac
constexpr int kMaxCep = 16;
constexpr int kMaxMel = 44;
constexpr int kStage = (kBatch - 1) * kFrameStep + kFrameSize;  // 2800 samples per batch
constexpr int kMelCols = 15;      // filter columns per wave; column 15 carries the wave's energy partial
constexpr int kMelSteps = 24;     // MFMA k-steps (4 bins each) per wave, upper bound
constexpr int kDctSteps = (kMaxMel + 3) / 4;

// LDS carve (floats).
constexpr int kSlot = 514;                           // spectrum row of frame f starts at f*514
constexpr int kXbufFloats = kBatch * kSlot;          // exchange buffers (fb*512) / spectrum rows; epilogue scratch
constexpr int kAbufFloats = 2816;                    // staged audio: kStage + up to 7 alignment samples
constexpr int kLogPitch = 48;                        // ln-mel row of a frame: nmel values, ln E at [nmel]
constexpr int kLbufFloats = kBatch * kLogPitch;
constexpr int kEpFloats = 4 * kBatch;                // energy partial of wave w for frame f at [w*16 + f]
constexpr int kHbufFloats = 2 * kHalo * kMaxCep;     // static features of the halo frames
constexpr int kSmemFloats = kXbufFloats + kAbufFloats + kLbufFloats + kEpFloats + kHbufFloats;

constexpr float kEps64 = 2.220446049250313e-16f;     // np.finfo(float).eps, mfcc-self5.py:131

// Output frames per tile such that the epilogue's two LDS arrays, (nt+8) and (nt+4) rows of
// numcep floats, fit in the exchange buffer; a multiple of the batch size.
inline int tile_frames_for(int numcep) {
  const int nt = (kXbufFloats / numcep - 12) / 2;
  return nt / kBatch * kBatch;
}

struct Tile {
  int32_t utt;  // utterance index
  int32_t t0;   // first output frame (utterance-relative)
  int32_t nt;   // number of output frames
  int32_t pad;
};

// How the [16 frames x 257 bins] x [257 x (nmel + energy)] filterbank product is cut for
// v_mfma_f32_16x16x4_f32: wave w owns filters first_filter[w] .. +nfilt[w]-1 in columns
// 0..14 of its 16x16 result and a partial energy sum in column 15, and walks kMelSteps 4-bin
// k-steps starting at bin kstart[w] (steps it does not need carry zero weights).
struct MelMap {
  int32_t kstart[4];
  int32_t first_filter[4];
  int32_t nfilt[4];
  int32_t ndct;  // k-steps of the DCT product = ceil(nmel / 4)
};

struct FusedParams {
  const void* audio;
  int64_t total_elems;        // elements in `audio` (bounds the 16-byte staging loads)
  const int64_t* sample_off;  // [n_utts + 1]
  const int64_t* frame_off;   // [n_utts + 1]
  const Tile* tiles;
  float* out;
  int64_t out_ld;
  const float* tw512;     // [512][2]  exp(-2 pi i m / 512)
  const float* melB;      // [4][kMelSteps][64] B operands (filter weights * spec_scale) in lane order
  const float* dctB;      // [kDctSteps][64] B operands of the DCT*lifter product in lane order
  const float* window;    // [512] or nullptr
  MelMap map;
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

typedef float f32x4 __attribute__((ext_vector_type(4)));

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

// Stage the samples of one 16-frame batch into LDS as float, pre-emphasised, zero beyond the
// utterance end (mfcc-self5.py:68 and :89-90).  Global memory is read in aligned 16-byte
// chunks; the LDS image keeps the chunk alignment, i.e. sample i of the batch sits at
// abuf[delta + i] where delta = (address of the batch's first sample) mod 16 bytes, in elements.
template <typename T>
__device__ __forceinline__ int stage_audio(const FusedParams& p, float* abuf, int64_t s_off, int64_t len, int tb,
                                           int tid) {
  constexpr int CE = 16 / (int)sizeof(T);  // elements per chunk
  const T* base = (const T*)p.audio;
  const int64_t g0 = s_off + (int64_t)tb * kFrameStep;  // first sample of the batch, in elements of `audio`
  const uintptr_t addr = (uintptr_t)(base + g0);
  const int delta = (int)((addr & 15) / sizeof(T));
  const T* a0 = (const T*)(addr & ~(uintptr_t)15);      // aligned chunk 0
  constexpr int kBig = 1 << 20;
  // chunk-relative element range that lies inside `audio`, and inside this utterance
  const int64_t lo64 = -(g0 - delta), hi64 = p.total_elems - (g0 - delta);
  const int e_lo = lo64 < -kBig ? -kBig : (int)lo64;
  const int e_hi = hi64 > kBig ? kBig : (int)hi64;
  const int64_t rem64 = len - (int64_t)tb * kFrameStep;  // samples of this utterance from the batch start on
  const int i_hi = delta + (rem64 > kBig ? kBig : (int)rem64);
  const int i_first = tb == 0 ? delta : -kBig;           // element whose predecessor is not part of the signal
  const int nchunk = (delta + kStage + CE - 1) / CE;
  const bool pre = p.use_preemph != 0;
  for (int c = tid; c < nchunk; c += 256) {
    const int e0 = c * CE;
    float x[CE + 1];
    if (e0 >= e_lo && e0 + CE <= e_hi) {
      const uint4 raw = *(const uint4*)(a0 + e0);
      if (sizeof(T) == 2) {
        const unsigned w[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          x[1 + 2 * j] = (float)(int16_t)(w[j] & 0xffff);
          x[2 + 2 * j] = (float)((int)w[j] >> 16);
        }
      } else {
        x[1] = __uint_as_float(raw.x); x[2] = __uint_as_float(raw.y);
        x[3] = __uint_as_float(raw.z); x[4] = __uint_as_float(raw.w);
      }
    } else {
#pragma unroll
      for (int j = 0; j < CE; ++j) x[1 + j] = (e0 + j >= e_lo && e0 + j < e_hi) ? (float)a0[e0 + j] : 0.f;
    }
    x[0] = (e0 - 1 >= e_lo && e0 - 1 < e_hi) ? (float)a0[e0 - 1] : 0.f;
    float y[CE];
#pragma unroll
    for (int j = 0; j < CE; ++j) {
      const int e = e0 + j;
      float v = x[j + 1];
      if (pre) {
        float d;
        if (sizeof(T) == 2) d = fmaf(-p.pre_lo, x[j], fmaf(-p.pre_hi, x[j], x[j + 1]));
        else d = __fsub_rn(x[j + 1], __fmul_rn(p.pre_hi, x[j]));  // numpy's float32 arithmetic
        v = e == i_first ? v : d;
      }
      y[j] = e < i_hi ? v : 0.f;  // elements below delta are never read
    }
#pragma unroll
    for (int j = 0; j < CE; j += 4) *(float4*)(abuf + e0 + j) = make_float4(y[j], y[j + 1], y[j + 2], y[j + 3]);
  }
  return delta;
}

// NCEP > 0 fixes numcep at compile time (cheap index arithmetic in the epilogue); 0 = runtime.
template <int OUT, int NCEP>
__global__ void __launch_bounds__(256, MFCC_MIN_WAVES) fused_kernel(const FusedParams p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* const xbuf = smem;
  float* const abuf = xbuf + kXbufFloats;
  float* const lbuf = abuf + kAbufFloats;
  float* const epbuf = lbuf + kLbufFloats;
  float* const hbuf = epbuf + kEpFloats;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = lane & 15;
  const int g4 = lane >> 4;
  const int fb = wave * 4 + g4;  // frame slot inside the batch

  const Tile tile = p.tiles[blockIdx.x];
  const int64_t s_off = p.sample_off[tile.utt];
  const int64_t len = p.sample_off[tile.utt + 1] - s_off;
  const int64_t f_off = p.frame_off[tile.utt];
  const int nframes = (int)(p.frame_off[tile.utt + 1] - f_off);
  const int t0 = tile.t0, nt = tile.nt;
  const int s_lo = (OUT == kOutMfcc39) ? max(0, t0 - kHalo) : t0;
  const int s_hi = (OUT == kOutMfcc39) ? min(nframes, t0 + nt + kHalo) : t0 + nt;
  const int nmel = p.nmel;
  const int ncep = NCEP > 0 ? NCEP : p.numcep;

  // per-lane twiddles: stage 1 uses W256^(q k1) = W512^(2 q k1), the untangle W512^(q + 16 s)
  float tw1r[16], tw1i[16], tw2r[8], tw2i[8];
#pragma unroll
  for (int k1 = 1; k1 < 16; ++k1) {
    const float2 w = *(const float2*)(p.tw512 + 2 * ((2 * q * k1) & 511));
    tw1r[k1] = w.x; tw1i[k1] = w.y;
  }
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    const float2 w = *(const float2*)(p.tw512 + 2 * (q + 16 * s));
    tw2r[s] = w.x; tw2i[s] = w.y;
  }
  const int partner_addr = ((lane & 48) | partner_lane(q)) << 2;
  const bool mag = p.spectrum != 0;

  auto stage = [&](int tb) -> int {
    return p.audio_dtype == 0 ? stage_audio<int16_t>(p, abuf, s_off, len, tb, tid)
                              : stage_audio<float>(p, abuf, s_off, len, tb, tid);
  };

  // columns nmel.. of the ln-mel rows are multiplied by zero DCT weights: keep them finite
  for (int i = tid; i < kLbufFloats; i += 256) lbuf[i] = 0.f;
  int delta = stage(s_lo);
  __syncthreads();

  int duty = 0;  // wave that runs the DCT of this batch
  for (int tb = s_lo; tb < s_hi; tb += kBatch, duty = (duty + 1) & 3) {
    // ---- FFT-512 of 4 frames per wave, 16 lanes per frame -----------------------------
    {
      const float* fa = abuf + delta + fb * kFrameStep;
      float xr[16], xi[16];
      if (delta & 1) {  // odd sample offset: the (x[2n], x[2n+1]) pairs are not 8-byte aligned in LDS
#pragma unroll
        for (int n1 = 0; n1 < 16; ++n1) {
          const int n = 16 * n1 + q;
          if (2 * (16 * n1) < kFrameSize) {
            const int nn = min(n, kFrameSize / 2 - 1);
            const bool ok = 2 * n + 1 < kFrameSize;
            const float a = fa[2 * nn], b = fa[2 * nn + 1];
            xr[n1] = ok ? a : 0.f; xi[n1] = ok ? b : 0.f;
          } else {
            xr[n1] = 0.f; xi[n1] = 0.f;
          }
        }
      } else {
        const float2* fa2 = (const float2*)__builtin_assume_aligned(fa, 8);
#pragma unroll
        for (int n1 = 0; n1 < 16; ++n1) {
          const int n = 16 * n1 + q;  // complex sample index: (x[2n], x[2n+1])
          if (2 * (16 * n1 + 15) + 1 < kFrameSize) {
            const float2 v = fa2[n];
            xr[n1] = v.x; xi[n1] = v.y;
          } else if (2 * (16 * n1) < kFrameSize) {
            const float2 v = fa2[min(n, kFrameSize / 2 - 1)];
            const bool ok = 2 * n + 1 < kFrameSize;
            xr[n1] = ok ? v.x : 0.f; xi[n1] = ok ? v.y : 0.f;
          } else {
            xr[n1] = 0.f; xi[n1] = 0.f;
          }
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

      float* const P = xbuf + fb * kSlot;
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        float br = lane_read(partner_addr, xr[15 - s]);
        float bi = lane_read(partner_addr, xi[15 - s]);
        if (q == 0) { br = xr[(16 - s) & 15]; bi = xi[(16 - s) & 15]; }
        float lo, hi;
        untangle_power(xr[s], xi[s], br, bi, tw2r[s], tw2i[s], lo, hi);
        if (mag) { lo = __builtin_amdgcn_sqrtf(lo); hi = __builtin_amdgcn_sqrtf(hi); }
        P[q + 16 * s] = lo;
        P[256 - q - 16 * s] = hi;
      }
      if (q == 0) {  // bin 128 pairs with itself: X[128] = conj Z[128]
        float m = 4.f * (xr[8] * xr[8] + xi[8] * xi[8]);
        if (mag) m = __builtin_amdgcn_sqrtf(m);
        P[128] = m;
        P[257] = 0.f; P[258] = 0.f; P[259] = 0.f;  // read (with zero weight) by the last MFMA k-step
      }
    }
    __syncthreads();  // spectrum rows complete; staged audio no longer needed

    if (OUT == kOutSpec || OUT == kOutLogspec) {
      const int nvalid = min(kBatch, s_hi - tb);
      for (int idx = tid; idx < nvalid * kNbin; idx += 256) {
        const int f = idx / kNbin, k = idx - f * kNbin;
        float v = xbuf[f * kSlot + k] * p.spec_scale;
        if (OUT == kOutLogspec) v = logf(v);
        p.out[(f_off + tb + f) * p.out_ld + k] = v;
      }
      if (tb + kBatch < s_hi) delta = stage(tb + kBatch);
      __syncthreads();
      continue;
    }

    // ---- mel filterbank on the matrix cores: D[frame][col] = sum_bin P[frame][bin] W[bin][col]
    {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      const float* Bw = p.melB + (wave * kMelSteps) * 64 + lane;
      // k-steps past a wave's range (zero weights) read finite leftovers of the exchange buffer
      const float* Prow = xbuf + q * kSlot + g4 + p.map.kstart[wave];
#pragma unroll
      for (int s = 0; s < kMelSteps; ++s)
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(Prow[4 * s], Bw[s * 64], acc, 0, 0, 0);
      // lane holds rows 4*g4 + r of column q: filters take ln at once (mfcc-self5.py:125-127,
      // unfloored by default: ln 0 = -inf), column 15 is this wave's share of the energy sum
      const int nf = p.map.nfilt[wave], ff = p.map.first_filter[wave];
      if (q < nf) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = acc[r];
          if (p.floor_mel && v == 0.f) v = kEps64;
          lbuf[(4 * g4 + r) * kLogPitch + ff + q] = logf(v);
        }
      } else if (q == 15) {
#pragma unroll
        for (int r = 0; r < 4; ++r) epbuf[wave * kBatch + 4 * g4 + r] = acc[r];
      }
    }
    if (tb + kBatch < s_hi) delta = stage(tb + kBatch);  // overlaps the other waves' MFMAs
    __syncthreads();  // ln-mel rows and energy partials complete; next batch staged

    if (OUT == kOutLogmel) {
      if (tid < kBatch) {
        float e = (epbuf[tid] + epbuf[16 + tid]) + (epbuf[32 + tid] + epbuf[48 + tid]);
        if (e == 0.f) e = kEps64;
        xbuf[tid] = logf(e);  // the exchange buffer is free until the next FFT
      }
      __syncthreads();
      const int nvalid = min(kBatch, s_hi - tb);
      const int w = nmel + 1;
      for (int idx = tid; idx < nvalid * w; idx += 256) {
        const int f = idx / w, j = idx - f * w;
        p.out[(f_off + tb + f) * p.out_ld + j] = j < nmel ? lbuf[f * kLogPitch + j] : xbuf[f];
      }
      __syncthreads();
      continue;
    }

    // ---- DCT-II * lifter on the matrix cores, c0 := ln E  (mfcc-self5.py:136-167) ----------
    // One wave per batch; the others go on to the next batch's FFT.  lbuf/epbuf are rewritten
    // only after the next spectrum barrier, which this wave reaches after it is done here.
    if (wave == duty) {
      const int nd = p.map.ndct;
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      const float* Lr = lbuf + q * kLogPitch + g4;
#pragma unroll
      for (int s = 0; s < kDctSteps; ++s)
        if (s < nd) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(Lr[4 * s], p.dctB[s * 64 + lane], acc, 0, 0, 0);
      if (q < ncep) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = 4 * g4 + r;
          const int t = tb + i;
          float v = acc[r];
          if (q == 0) {  // mfcc-self5.py:130-132 and :164
            float e = (epbuf[i] + epbuf[16 + i]) + (epbuf[32 + i] + epbuf[48 + i]);
            if (e == 0.f) e = kEps64;
            v = logf(e);
          }
          if (t < s_hi) {
            if (t >= t0 && t < t0 + nt) p.out[(f_off + t) * p.out_ld + q] = v;
            else hbuf[(t < t0 ? t - s_lo : kHalo + t - (t0 + nt)) * kMaxCep + q] = v;
          }
        }
      }
    }
  }

  if (OUT == kOutMfcc39) {
    // ---- epilogue: statics back from L2, delta, delta-delta, remaining columns out ----------
    __syncthreads();  // all static columns of the tile are written and visible to this workgroup
    float* const sb = xbuf;                             // [(s_hi - s_lo)][ncep]
    const int d_lo = max(0, t0 - 2), d_hi = min(nframes, t0 + nt + 2);
    float* const db = xbuf + (s_hi - s_lo) * ncep;      // [(d_hi - d_lo)][ncep]
    const int last = nframes - 1;
    for (int idx = tid; idx < (s_hi - s_lo) * ncep; idx += 256) {
      const int r = idx / ncep, c = idx - r * ncep;
      const int t = s_lo + r;
      float v;
      if (t < t0) v = hbuf[r * kMaxCep + c];
      else if (t >= t0 + nt) v = hbuf[(kHalo + t - (t0 + nt)) * kMaxCep + c];
      else v = p.out[(f_off + t) * p.out_ld + c];
      sb[idx] = v;
    }
    __syncthreads();
    for (int idx = tid; idx < (d_hi - d_lo) * ncep; idx += 256) {
      const int r = idx / ncep, c = idx - r * ncep;
      const int t = d_lo + r;
      const float a = sb[(max(t - 2, 0) - s_lo) * ncep + c];
      const float b = sb[(max(t - 1, 0) - s_lo) * ncep + c];
      const float m = sb[(t - s_lo) * ncep + c];
      const float d = sb[(min(t + 1, last) - s_lo) * ncep + c];
      const float e = sb[(min(t + 2, last) - s_lo) * ncep + c];
      // (-2a - b + 0m + d + 2e) / 10; the zero-weight centre tap keeps NaN propagation
      // identical to np.dot over the 5-row window (mfcc-self5.py:177-178)
      db[idx] = (2.f * (e - a) + (d - b) + 0.f * m) * 0.1f;
    }
    __syncthreads();
    const int width = 2 * ncep;
    for (int idx = tid; idx < nt * width; idx += 256) {
      const int r = idx / width, c = idx - r * width;
      const int t = t0 + r;
      float v;
      if (c < ncep) {
        v = db[(t - d_lo) * ncep + c];
      } else {
        const int cc = c - ncep;
        const float a = db[(max(t - 2, 0) - d_lo) * ncep + cc];
        const float b = db[(max(t - 1, 0) - d_lo) * ncep + cc];
        const float m = db[(t - d_lo) * ncep + cc];
        const float d = db[(min(t + 1, last) - d_lo) * ncep + cc];
        const float e = db[(min(t + 2, last) - d_lo) * ncep + cc];
        v = (2.f * (e - a) + (d - b) + 0.f * m) * 0.1f;
      }
      p.out[(f_off + t) * p.out_ld + ncep + c] = v;
    }
  }
}

}  // namespace mfcc
