// General-geometry MFCC kernel: the same pipeline as fused_kernel.h for every framing the
// reference's winfft can derive (frame_size = int(fs * frame_time / 1000), num_fft = next power of
// two, mfcc-self5.py:75-84), any pad_fac of delta (:169-179) and filterbank / cepstrum sizes the
// specialised kernel has no template for.  The 400/160/512 and 200/80/512 framings of the
// reference's own scripts run in fused_kernel.h, about three times faster per frame.
//
// Work decomposition
//   grid       one workgroup (4 waves) per tile of consecutive output frames of one utterance
//              (same Tile list as the specialised kernel, shorter tiles);
//   wave       FFT lengths up to 512 (general16_kernel): four frames at a time, 16 lanes per frame, on the
//              register transform of fft_lane.h -- an nfft-point transform of a frame of at most nfft
//              samples is every (512 / nfft)-th bin of the 512-point transform of the same zero-padded
//              frame; the samples come straight from global memory, everything after the transform
//              (filters, ln, DCT) runs on the frame's 16 lanes out of the wave's own LDS rows, so the
//              waves of a workgroup never wait for each other until the tile's rows are complete.
//              1024 and 2048 points (generalw_kernel): L = nfft / 32 lanes per frame (two frames or one per
//              wave).  The frame is packed as M = nfft / 2 complex points z[n] = x[2n] + i x[2n+1]; M = 16 * 16 * R
//              (R = 2 or 4) runs as three passes -- two 16-point register transforms of fft_lane.h and an
//              R-point one -- with twiddle multiplications and an exchange through the wave's own LDS buffer
//              between them (a wave's LDS operations execute in program order, so no workgroup barrier is
//              needed), then the untangle into the M + 1 spectrum values, the sparse triangular filters
//              (the frame's lanes share the filters and, when there are fewer filters than half the lanes,
//              each filter's band), ln, DCT-II * lifter (one lane per coefficient);
//   tile end   the static rows of the tile and its halo (2 * pad_fac frames on each side that
//              continues) sit in LDS: delta over them, then delta of the deltas, both with the
//              indices clamped to the utterance (np.pad(mode='edge'), mfcc-self5.py:174).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fused_kernel.h"

namespace mfcc {

constexpr int kGenMaxFft = 2048;   // largest FFT (frame_size <= 2048: 25 ms up to 81.9 kHz)
constexpr int kGenMaxMel = 128;
constexpr int kGenMaxCep = 64;
constexpr int kGenMaxPad = 16;     // pad_fac of delta
constexpr int kGenMaxTile = 128;   // output frames per tile at most

// Constant tables of a general plan (device memory).
struct GeneralConst {
  const float* tw;        // [nfft/2][2]  exp(-2 pi i k / nfft)
  const float* bankc;     // triangular weights (mfcc-self5.py:102-117), compact: only the bins of each filter's band
  const int32_t* band;    // [nmel][3]    first bin with a non-zero weight, one past the last, offset of the band in bankc
  const float* dctl;      // [nmel][numcep] DCT-II 'ortho' * lifter (mfcc-self5.py:136-156)
  const float* window;    // [frame_size] or nullptr
  int32_t frame_size, frame_step, nfft, log2m;  // log2m = log2(nfft / 2)
  int32_t used;           // samples of a frame that enter the FFT = min(frame_size, nfft) (np.fft.rfft truncates)
  int32_t nmel, numcep, pad_fac;
  int32_t spectrum, floor_mel;
  int32_t keep_c0;        // 1: coefficient 0 stays the DCT's own value instead of ln E
  float pre_hi, pre_lo;   // 0, 0 = no pre-emphasis
  float inv_nfft;         // power spectrum scale 1/nfft (mfcc-self5.py:97)
  float log_floor;
  float delta_denom;      // 2 * sum i^2, mfcc-self5.py:172
  int32_t stat_rows;      // rows of the statics / delta arrays in LDS (>= tile frames + 4 * pad_fac)
  int32_t bankc_lds;      // > 0: floats of LDS that hold a copy of bankc (read from there)
  int32_t dct_lds;        // > 0: floats of LDS that hold a copy of dctl
  int32_t lane16;         // 1: nfft <= 512, general16_kernel runs; tw then holds [512][2] exp(-2 pi i k / 512)
};

// dynamic LDS of the kernels, in floats: the twiddle table, the filters' band table, per wave the frames'
// exchange buffers and ln-mel rows, the filter weights and the DCT matrix when the plan found room for them
// (bankc_lds, dct_lds floats; not counted by general_smem_bytes); per workgroup the static and delta rows of
// a tile (MFCC-39 only)
// general16_kernel: the 512-point twiddle table (1024 floats), the filters' band table (3 ints per filter),
// per wave four 2 KiB exchange buffers (the frames' spectrum rows afterwards) and four ln-mel rows
__host__ __device__ inline int general16_lrow_floats(int nmel) { return (nmel + 1 + 3) & ~3; }
__host__ __device__ inline int general16_wave_floats(int nmel) { return 4 * 512 + 4 * general16_lrow_floats(nmel); }
__host__ __device__ inline int general16_band_floats(int nmel) { return (3 * nmel + 3) & ~3; }
// generalw_kernel (1024 and 2048 points, L = nfft / 32 lanes per frame, 64 / L frames per wave): the twiddle table
// exp(-2 pi i k / nfft), k < nfft / 2 (nfft floats), the band table, per wave the frames' exchange buffers -- 16 rows of
// L + L / 16 complex values each, so that the column reads of the second pass spread over the banks -- and ln-mel rows
__host__ __device__ inline int generalw_buf_floats(int nfft) { return 2 * 16 * (nfft / 32 + nfft / 512); }  // one frame
__host__ __device__ inline int generalw_wave_floats(int nfft, int nmel) {
  const int fpw = 64 / (nfft / 32);
  return fpw * generalw_buf_floats(nfft) + fpw * general16_lrow_floats(nmel);
}
// LDS floats of a kernel apart from the optional table copies and the static / delta rows
inline size_t general_fixed_floats(int nfft, int nmel) {
  return nfft <= 512 ? (size_t)1024 + general16_band_floats(nmel) + (size_t)4 * general16_wave_floats(nmel)
                     : (size_t)nfft + general16_band_floats(nmel) + (size_t)4 * generalw_wave_floats(nfft, nmel);
}
inline size_t general_smem_bytes(int nfft, int nmel, int numcep, int stat_rows) {
  return (general_fixed_floats(nfft, nmel) + (size_t)2 * stat_rows * numcep) * sizeof(float);
}

// Tile end of both kernels: delta (mfcc-self5.py:169-179) of rows d_lo .. d_hi-1 of the static rows in LDS,
// then delta of the deltas for the tile's own rows; both clamp their row indices to the utterance; rows out.
__device__ __forceinline__ void general_tile_rows(const GeneralConst& gc, const float* stat, float* dlt, int s_lo, int t0, int nt, int nframes,
                                                  float* out_rows, int64_t ld, int tid) {
  const int ncep = gc.numcep, pf = gc.pad_fac;
  __syncthreads();
  const int d_lo = max(0, t0 - pf), d_hi = min(nframes, t0 + nt + pf);
  const float denom = gc.delta_denom;
  for (int i = tid; i < (d_hi - d_lo) * ncep; i += 256) {
    const int r = d_lo + i / ncep, c = i % ncep;
    float acc = 0.f;
    for (int k = -pf; k <= pf; ++k) {
      const int rr = min(max(r + k, 0), nframes - 1);
      acc = fmaf((float)k, stat[(rr - s_lo) * ncep + c], acc);
    }
    dlt[(r - s_lo) * ncep + c] = acc / denom;
  }
  __syncthreads();
  for (int i = tid; i < nt * ncep; i += 256) {
    const int r = t0 + i / ncep, c = i % ncep;
    float acc = 0.f;
    for (int k = -pf; k <= pf; ++k) {
      const int rr = min(max(r + k, 0), nframes - 1);
      acc = fmaf((float)k, dlt[(rr - s_lo) * ncep + c], acc);
    }
    float* const o = out_rows + (int64_t)r * ld;   // out_rows = row 0 of the utterance
    o[c] = stat[(r - s_lo) * ncep + c];
    o[ncep + c] = dlt[(r - s_lo) * ncep + c];
    o[2 * ncep + c] = acc / denom;
  }
}

// TABLES_IN_LDS (both kernels): the plan found room for the filter weights and the DCT matrix in LDS (both or
// neither); a compile-time choice so that their reads are LDS or global instructions, not flat ones.

// ---- FFT lengths up to 512: four frames per wave ------------------------------------------------------------
template <typename T, bool TABLES_IN_LDS>
__global__ void __launch_bounds__(256, 3) general16_kernel(const FusedParams p, const GeneralConst* __restrict__ gcp, int out_kind) {
  extern __shared__ __attribute__((aligned(16))) float gsm[];
  const GeneralConst gc = *gcp;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q_lane = lane & 15, g4 = lane >> 4;
  const int nbin = (gc.nfft >> 1) + 1;
  const int dshift = 8 - gc.log2m;            // log2(512 / nfft): bin j of the nfft-point transform = bin j << dshift of the 512-point one
  const int dmask = (1 << dshift) - 1;
  const int nmel = gc.nmel, ncep = gc.numcep, pf = gc.pad_fac;
  const cf* const twl = (const cf*)gsm;       // [512] exp(-2 pi i m / 512)
  float* const tables = gsm + 1024;           // [bankc_lds] filter weights, [dct_lds] DCT * lifter
  typedef const float __attribute__((address_space(3))) * lds_cptr;
  typedef const float __attribute__((address_space(1))) * glb_cptr;
  typedef typename std::conditional<TABLES_IN_LDS, lds_cptr, glb_cptr>::type tab_cptr;
  const tab_cptr bankw = TABLES_IN_LDS ? (tab_cptr)(uintptr_t)(lds_cptr)tables : (tab_cptr)(uintptr_t)gc.bankc;
  const tab_cptr dctw = TABLES_IN_LDS ? (tab_cptr)(uintptr_t)(lds_cptr)(tables + gc.bankc_lds) : (tab_cptr)(uintptr_t)gc.dctl;
  int32_t* const bandl = (int32_t*)(tables + gc.bankc_lds + gc.dct_lds);  // [nmel][3]
  const int lpitch = general16_lrow_floats(nmel);
  float* const wbase = (float*)bandl + general16_band_floats(nmel) + wave * general16_wave_floats(nmel);
  float* const xch = wbase + g4 * 512;                          // this frame's exchange buffer, its spectrum row afterwards
  float* const lrow = wbase + 4 * 512 + g4 * lpitch;            // [nmel] ln mel of this frame
  float* const stat = (float*)bandl + general16_band_floats(nmel) + 4 * general16_wave_floats(nmel);  // [stat_rows][ncep]
  float* const dlt = stat + gc.stat_rows * ncep;

  const Tile tile = p.tiles[blockIdx.x];
  const int64_t s_off = p.sample_off[tile.utt];
  const int64_t len = p.sample_len ? p.sample_len[tile.utt] : p.sample_off[tile.utt + 1] - s_off;
  const int64_t f_off = p.frame_off[tile.utt];
  const int nframes = (int)(p.frame_off[tile.utt + 1] - f_off);
  const int t0 = tile.t0, nt = tile.nt;
  const bool full = out_kind == kOutMfcc39;
  const int s_lo = full ? max(0, t0 - 2 * pf) : t0;
  const int s_hi = full ? min(nframes, t0 + nt + 2 * pf) : t0 + nt;
  const T* const x = (const T*)p.audio + s_off;
  const int64_t ld = p.out_ld;
  const bool mag = gc.spectrum != 0;
  const bool pre_on = !(gc.pre_hi == 0.f && gc.pre_lo == 0.f);

  for (int i = tid; i < 1024; i += 256) gsm[i] = gc.tw[i];
  for (int i = tid; i < gc.bankc_lds; i += 256) tables[i] = gc.bankc[i];  // the plan pads bankc to this length
  for (int i = tid; i < gc.dct_lds; i += 256) tables[gc.bankc_lds + i] = i < nmel * ncep ? gc.dctl[i] : 0.f;
  for (int i = tid; i < 3 * nmel; i += 256) bandl[i] = gc.band[i];
  __syncthreads();

  for (int tq = s_lo + 4 * wave; tq < s_hi; tq += 16) {
    // The lane's index inside its frame passes through an empty asm statement: everything derived from it below
    // (sample offsets, window addresses, bin tests, twiddle addresses) is then computed where it is used -- hoisted
    // out of this loop, as the compiler would do with values that do not change, they need more registers than there are.
    int q = q_lane;
    asm volatile("" : "+v"(q));
    const int t = tq + g4;
    const bool live = t < s_hi;  // frame slots past the tile's last frame transform zeros and store nothing
    // ---- framing (mfcc-self5.py:85-95): samples t*step .. +used, zeros past the utterance end; point n = 16 n1 + q
    //      of the packed frame z[n] = y[2n] + i y[2n+1].  Indices are clamped into the utterance so that every
    //      load is legal; what lies outside is zeroed.
    // 32-bit sample positions relative to the frame's first sample: lo_r = the utterance's first sample, hi_r = its last
    const int64_t g0 = (int64_t)(live ? t : s_lo) * gc.frame_step;
    const T* const xf = x + g0;
    constexpr int64_t kFar = 1 << 29;
    const int lo_r = (int)(g0 < kFar ? -g0 : -kFar), hi_r = (int)(len - 1 - g0 < kFar ? len - 1 - g0 : kFar);
    cf xr[16];
    // all loads of the frame first (one memory round trip per four frames; the kernel runs few waves per SIMD,
    // so registers are not what limits it); registers beyond the frame's samples load clamped addresses and are zeroed
    T rm[16], ra[16], rb[16];
#pragma unroll
    for (int n1 = 0; n1 < 16; ++n1) {
      const int i = 2 * (16 * n1 + q);
      rm[n1] = xf[min(max(i - 1, lo_r), hi_r)];
      ra[n1] = xf[min(i, hi_r)];
      rb[n1] = xf[min(i + 1, hi_r)];
    }
#pragma unroll
    for (int n1 = 0; n1 < 16; ++n1) {
      const int i = 2 * (16 * n1 + q);
      const float xm = (float)rm[n1], xa = (float)ra[n1], xb = (float)rb[n1];
      // y[0] = x[0], y[n] = x[n] - c x[n-1] (mfcc-self5.py:68), two-term coefficient for int16 samples
      auto emph = [&](float cur, float prev, bool first) {
        if (first || !pre_on) return cur;
        if (sizeof(T) == 2) return fmaf(-gc.pre_lo, prev, fmaf(-gc.pre_hi, prev, cur));
        return __fsub_rn(cur, __fmul_rn(gc.pre_hi, prev));
      };
      float y0 = (live && i < gc.used && i <= hi_r) ? emph(xa, xm, i == lo_r) : 0.f;
      float y1 = (live && i + 1 < gc.used && i + 1 <= hi_r) ? emph(xb, xa, false) : 0.f;
      if (gc.window) {
        y0 *= gc.window[min(i, gc.frame_size - 1)];
        y1 *= gc.window[min(i + 1, gc.frame_size - 1)];
      }
      xr[n1] = mk(y0, y1);
    }
    // ---- 512-point transform of the zero-padded frame on 16 lanes (fft_lane.h)
    fft16(xr);
    cf* const xc = (cf*)xch;
#pragma unroll
    for (int k1 = 0; k1 < 16; ++k1) {
      if (k1) xr[k1] = cmul(xr[k1], twl[(2 * q * k1) & 511]);  // W256^(q k1)
      xc[xpose_index(k1, q)] = xr[k1];
    }
    wave_sync();
#pragma unroll
    for (int n2 = 0; n2 < 16; ++n2) xr[n2] = xc[xpose_index(q, n2)];
    fft16(xr);
    wave_sync();  // every lane has read its row before the spectrum row overwrites the buffer
    // ---- untangle into the frame's spectrum row: value |X|^2 / nfft or |X| (mfcc-self5.py:96-97) of every
    //      (512 / nfft)-th bin
    float* const orow = (out_kind == kOutSpec || out_kind == kOutLogspec) && live && t >= t0 ? p.out + (f_off + t) * ld : nullptr;
    auto put = [&](int k, float p4) {
      if (k & dmask) return;
      const float v = mag ? 0.5f * __builtin_amdgcn_sqrtf(p4) : p4 * (0.25f * gc.inv_nfft);
      xch[k >> dshift] = v;
      if (orow) orow[k >> dshift] = out_kind == kOutLogspec ? fast_ln(fmaxf(v, gc.log_floor)) : v;
    };
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      cf pb = mk(partner_dpp(xr[15 - s].x), partner_dpp(xr[15 - s].y));  // Z[256 - k] sits in lane (16 - q) & 15 of the frame's row
      if (q == 0) pb = xr[(16 - s) & 15];  // lane 0 pairs inside itself
      float lo, hi2;
      untangle_power(xr[s], pb, twl[q + 16 * s], lo, hi2);
      put(q + 16 * s, lo);
      put(256 - q - 16 * s, hi2);
    }
    if (q == 0) put(128, 4.f * norm2(xr[8]));
    wave_sync();
    if (out_kind == kOutSpec || out_kind == kOutLogspec) continue;
    // ---- mel energies and frame energy (mfcc-self5.py:119-134) on the frame's 16 lanes: ln of the unfloored band
    //      sums (-inf on an empty band) unless floor_mel; ln E floored at the float64 epsilon
    float e = 0.f;
    for (int k = q; k < nbin; k += 16) e += xch[k];
    e += row_ror<8>(e);
    e += row_ror<4>(e);
    e += row_ror<2>(e);
    e += row_ror<1>(e);
    if (e == 0.f) e = kEps64;
    const float lne = fast_ln(e);
    for (int j = q; j < nmel; j += 16) {
      const int lo = bandl[3 * j], hi3 = bandl[3 * j + 1];
      const tab_cptr w = bankw + bandl[3 * j + 2] - lo;  // w[k] = weight of bin k, lo <= k < hi
      float acc = 0.f;
      int k = lo;
      for (; k + 4 <= hi3; k += 4) {  // four weights in flight; the sum keeps the order of the bins
        const float w0 = w[k], w1 = w[k + 1], w2 = w[k + 2], w3 = w[k + 3];
        acc = fmaf(xch[k], w0, acc);
        acc = fmaf(xch[k + 1], w1, acc);
        acc = fmaf(xch[k + 2], w2, acc);
        acc = fmaf(xch[k + 3], w3, acc);
      }
      for (; k < hi3; ++k) acc = fmaf(xch[k], w[k], acc);
      if (gc.floor_mel && acc == 0.f) acc = kEps64;
      lrow[j] = fast_ln(acc);
    }
    wave_sync();
    const bool own = live && t >= t0 && t < t0 + nt;
    if (out_kind == kOutLogmel) {
      if (own) {
        float* const o = p.out + (f_off + t) * ld;
        for (int j = q; j < nmel; j += 16) o[j] = lrow[j];
        if (q == 0) o[nmel] = lne;
      }
      wave_sync();
      continue;
    }
    // ---- DCT-II * lifter, c0 := ln E (mfcc-self5.py:136-167)
    for (int c = q; c < ncep; c += 16) {
      float acc = 0.f;
      int n = 0;
      for (; n + 4 <= nmel; n += 4) {
        const float w0 = dctw[n * ncep + c], w1 = dctw[(n + 1) * ncep + c], w2 = dctw[(n + 2) * ncep + c], w3 = dctw[(n + 3) * ncep + c];
        acc = fmaf(lrow[n], w0, acc);
        acc = fmaf(lrow[n + 1], w1, acc);
        acc = fmaf(lrow[n + 2], w2, acc);
        acc = fmaf(lrow[n + 3], w3, acc);
      }
      for (; n < nmel; ++n) acc = fmaf(lrow[n], dctw[n * ncep + c], acc);
      if (c == 0 && !gc.keep_c0) acc = lne;
      if (full) {
        if (live) stat[(t - s_lo) * ncep + c] = acc;
      } else if (own) {
        p.out[(f_off + t) * ld + c] = acc;
      }
    }
    wave_sync();
  }
  if (!full) return;
  general_tile_rows(gc, stat, dlt, s_lo, t0, nt, nframes, p.out + f_off * ld, ld, tid);
}

// ---- 1024 and 2048 points: L = 32 or 64 lanes per frame (R = L / 16 = 2 or 4), 64 / L frames per wave ----------------
// The nfft / 2 = 16 L packed complex points of a frame, n = L n1 + n2, k = k1 + 16 (k2a + 16 k2b):
//   pass 1   lane n2:          Y[k1] = W_M^(n2 k1) * sum_n1 z[L n1 + n2] W16^(n1 k1)               (16-point, registers)
//   exchange                    lane (k1, r) takes Y[k1][n2 = R m + r], m = 0..15
//   pass 2   lane R k1 + r:     U[k2a] = W_L^(r k2a) * sum_m Y[k1][R m + r] W16^(m k2a)             (16-point, registers)
//   exchange                    every lane takes 16 / R groups of the R values U_r[k2a] of one (k1, k2a)
//   pass 3                      Z[k1 + 16 k2a + 256 k2b] = sum_r U_r[k2a] W_R^(r k2b)                (R-point, registers)
//   Z to LDS in natural order; untangle in pairs (p, M - p), p = lane + L j, as in fft_lane.h.
template <typename T, bool TABLES_IN_LDS, int R>
__global__ void __launch_bounds__(256, 3) generalw_kernel(const FusedParams p, const GeneralConst* __restrict__ gcp, int out_kind) {
  constexpr int L = 16 * R, M = 16 * L, FPW = 64 / L;
  // complex values per row of the first exchange and per k1 block of the second: R lanes read or write R
  // neighbouring values, the next R lanes the same R values of the next row -- with rows 2 R floats past a
  // multiple of 64 banks the 32 lanes of an access cycle cover the 64 banks once
  constexpr int P1 = L + R;
  constexpr int P2 = 17 * R;
  static_assert(16 * P2 <= 16 * P1 && M <= 16 * P1, "every stage fits the frame's buffer");
  extern __shared__ __attribute__((aligned(16))) float gsm[];
  const GeneralConst gc = *gcp;      // gc.nfft == 2 M
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li_lane = lane & (L - 1), fw = lane / L;
  const int nbin = M + 1;
  const int nmel = gc.nmel, ncep = gc.numcep, pf = gc.pad_fac;
  const cf* const twl = (const cf*)gsm;       // [M] exp(-2 pi i k / nfft)
  float* const tables = gsm + 2 * M;          // [bankc_lds] filter weights, [dct_lds] DCT * lifter
  typedef const float __attribute__((address_space(3))) * lds_cptr;
  typedef const float __attribute__((address_space(1))) * glb_cptr;
  typedef typename std::conditional<TABLES_IN_LDS, lds_cptr, glb_cptr>::type tab_cptr;
  const tab_cptr bankw = TABLES_IN_LDS ? (tab_cptr)(uintptr_t)(lds_cptr)tables : (tab_cptr)(uintptr_t)gc.bankc;
  const tab_cptr dctw = TABLES_IN_LDS ? (tab_cptr)(uintptr_t)(lds_cptr)(tables + gc.bankc_lds) : (tab_cptr)(uintptr_t)gc.dctl;
  int32_t* const bandl = (int32_t*)(tables + gc.bankc_lds + gc.dct_lds);  // [nmel][3]
  const int lpitch = general16_lrow_floats(nmel);
  const int nmel_pad = nmel > 1 ? 1 << (32 - __builtin_clz(nmel - 1)) : 1;  // next power of two
  float* const wbase = (float*)bandl + general16_band_floats(nmel) + wave * generalw_wave_floats(2 * M, nmel);
  float* const buf = wbase + fw * generalw_buf_floats(2 * M);            // this frame's exchange buffer, its spectrum row afterwards
  cf* const bc = (cf*)buf;
  float* const lrow = wbase + FPW * generalw_buf_floats(2 * M) + fw * lpitch;
  float* const stat = (float*)bandl + general16_band_floats(nmel) + 4 * generalw_wave_floats(2 * M, nmel);  // [stat_rows][ncep]
  float* const dlt = stat + gc.stat_rows * ncep;

  const Tile tile = p.tiles[blockIdx.x];
  const int64_t s_off = p.sample_off[tile.utt];
  const int64_t len = p.sample_len ? p.sample_len[tile.utt] : p.sample_off[tile.utt + 1] - s_off;
  const int64_t f_off = p.frame_off[tile.utt];
  const int nframes = (int)(p.frame_off[tile.utt + 1] - f_off);
  const int t0 = tile.t0, nt = tile.nt;
  const bool full = out_kind == kOutMfcc39;
  const int s_lo = full ? max(0, t0 - 2 * pf) : t0;
  const int s_hi = full ? min(nframes, t0 + nt + 2 * pf) : t0 + nt;
  const T* const x = (const T*)p.audio + s_off;
  const int64_t ld = p.out_ld;
  const bool mag = gc.spectrum != 0;
  const bool pre_on = !(gc.pre_hi == 0.f && gc.pre_lo == 0.f);

  for (int i = tid; i < 2 * M; i += 256) gsm[i] = gc.tw[i];
  for (int i = tid; i < gc.bankc_lds; i += 256) tables[i] = gc.bankc[i];  // the plan pads bankc to this length
  for (int i = tid; i < gc.dct_lds; i += 256) tables[gc.bankc_lds + i] = i < nmel * ncep ? gc.dctl[i] : 0.f;
  for (int i = tid; i < 3 * nmel; i += 256) bandl[i] = gc.band[i];
  __syncthreads();
  // W_M^x = exp(-2 pi i 2x / nfft), 0 <= x < M, from the half table: the second half of the circle is minus the first
  auto w_m = [&](int xx) {
    const int i2 = 2 * xx;
    const cf w = twl[i2 & (M - 1)];
    return i2 < M ? w : mk(-w.x, -w.y);
  };
  // twiddles of this lane, kept for all its frames: pass 1 W_M^(n2 k1), pass 2 W_L^(r k2a) = W_M^(16 r k2a)
  cf tw1[16], tw2[16];
#pragma unroll
  for (int k = 1; k < 16; ++k) {
    tw1[k] = w_m(li_lane * k);
    tw2[k] = w_m(16 * (li_lane % R) * k);
  }

  for (int tq = s_lo + FPW * wave; tq < s_hi; tq += 4 * FPW) {
    // (the lane's index passes through an empty asm statement so that what is derived from it is computed here
    // rather than hoisted out of the loop into more registers than there are, as in general16_kernel)
    int li = li_lane;
    asm volatile("" : "+v"(li));
    const int t = tq + fw;
    const bool live = t < s_hi;
    const int64_t g0 = (int64_t)(live ? t : s_lo) * gc.frame_step;
    const T* const xf = x + g0;
    constexpr int64_t kFar = 1 << 29;
    const int lo_r = (int)(g0 < kFar ? -g0 : -kFar), hi_r = (int)(len - 1 - g0 < kFar ? len - 1 - g0 : kFar);
    cf xr[16];
    T rm[16], ra[16], rb[16];
#pragma unroll
    for (int n1 = 0; n1 < 16; ++n1) {
      const int i = 2 * (L * n1 + li);
      rm[n1] = xf[min(max(i - 1, lo_r), hi_r)];
      ra[n1] = xf[min(i, hi_r)];
      rb[n1] = xf[min(i + 1, hi_r)];
    }
#pragma unroll
    for (int n1 = 0; n1 < 16; ++n1) {
      const int i = 2 * (L * n1 + li);
      const float xm = (float)rm[n1], xa = (float)ra[n1], xb = (float)rb[n1];
      auto emph = [&](float cur, float prev, bool first) {
        if (first || !pre_on) return cur;
        if (sizeof(T) == 2) return fmaf(-gc.pre_lo, prev, fmaf(-gc.pre_hi, prev, cur));
        return __fsub_rn(cur, __fmul_rn(gc.pre_hi, prev));
      };
      float y0 = (live && i < gc.used && i <= hi_r) ? emph(xa, xm, i == lo_r) : 0.f;
      float y1 = (live && i + 1 < gc.used && i + 1 <= hi_r) ? emph(xb, xa, false) : 0.f;
      if (gc.window) {
        y0 *= gc.window[min(i, gc.frame_size - 1)];
        y1 *= gc.window[min(i + 1, gc.frame_size - 1)];
      }
      xr[n1] = mk(y0, y1);
    }
    // ---- pass 1
    fft16(xr);
#pragma unroll
    for (int k1 = 0; k1 < 16; ++k1) {
      if (k1) xr[k1] = cmul(xr[k1], tw1[k1]);
      bc[k1 * P1 + li] = xr[k1];
    }
    wave_sync();
    // ---- pass 2: lane R k1 + r
    const int k1b = li / R, rr = li % R;
#pragma unroll
    for (int m = 0; m < 16; ++m) xr[m] = bc[k1b * P1 + R * m + rr];
    fft16(xr);
    wave_sync();  // every lane has read its column before the buffer is written again
#pragma unroll
    for (int k2a = 0; k2a < 16; ++k2a) {
      if (k2a) xr[k2a] = cmul(xr[k2a], tw2[k2a]);
      bc[k1b * P2 + k2a * R + rr] = xr[k2a];
    }
    wave_sync();
    // ---- pass 3: 16 / R groups of R values per lane; group j = (k1, k2a) = (j / 16, j % 16)
#pragma unroll
    for (int u = 0; u < 16 / R; ++u) {
      const int j = li * (16 / R) + u;
#pragma unroll
      for (int r2 = 0; r2 < R; ++r2) xr[u * R + r2] = bc[(j >> 4) * P2 + (j & 15) * R + r2];
    }
    wave_sync();
#pragma unroll
    for (int u = 0; u < 16 / R; ++u) {
      const int j = li * (16 / R) + u;
      const int kb = (j >> 4) + 16 * (j & 15);
      if (R == 2) {
        const cf a = xr[2 * u], b = xr[2 * u + 1];
        bc[kb] = a + b;
        bc[kb + 256] = a - b;
      } else {
        radix4(xr[4 * u], xr[4 * u + 1], xr[4 * u + 2], xr[4 * u + 3]);
#pragma unroll
        for (int b = 0; b < 4; ++b) bc[kb + 256 * b] = xr[4 * u + b];
      }
    }
    wave_sync();
    // ---- untangle (fft_lane.h): pairs (pp, M - pp), pp = li + L j < M / 2; pp = 0 pairs Z[0] with itself and
    //      yields bins 0 and M; bin M / 2 is its own partner
    cf za[8], zb[8], zw[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int pp = li + L * j;
      za[j] = bc[pp];
      zb[j] = bc[(M - pp) & (M - 1)];
      zw[j] = twl[pp];
    }
    const cf zmid = bc[M / 2];
    wave_sync();  // all of Z is in registers: the buffer becomes the frame's spectrum row
    float* const orow = (out_kind == kOutSpec || out_kind == kOutLogspec) && live && t >= t0 ? p.out + (f_off + t) * ld : nullptr;
    auto put = [&](int k, float p4) {
      const float v = mag ? 0.5f * __builtin_amdgcn_sqrtf(p4) : p4 * (0.25f * gc.inv_nfft);
      buf[k] = v;
      if (orow) orow[k] = out_kind == kOutLogspec ? fast_ln(fmaxf(v, gc.log_floor)) : v;
    };
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int pp = li + L * j;
      float lo, hi2;
      untangle_power(za[j], zb[j], zw[j], lo, hi2);
      put(pp, lo);
      put(M - pp, hi2);
    }
    if (li == 0) put(M / 2, 4.f * norm2(zmid));
    wave_sync();
    if (out_kind == kOutSpec || out_kind == kOutLogspec) continue;
    // ---- mel energies and frame energy (mfcc-self5.py:119-134) on the frame's L lanes
    float e = 0.f;
    for (int k = li; k < nbin; k += L) e += buf[k];
#pragma unroll
    for (int m = L / 2; m >= 1; m >>= 1) e += __shfl_xor(e, m, L);
    if (e == 0.f) e = kEps64;
    const float lne = fast_ln(e);
    // a band's sum over bins [a, b): four weights in flight; the sum keeps the order of the bins
    auto band_sum = [&](int j, int a, int b) {
      const tab_cptr w = bankw + bandl[3 * j + 2] - bandl[3 * j];  // w[k] = weight of bin k of filter j
      float acc = 0.f;
      int k = a;
      for (; k + 4 <= b; k += 4) {
        const float w0 = w[k], w1 = w[k + 1], w2 = w[k + 2], w3 = w[k + 3];
        acc = fmaf(buf[k], w0, acc);
        acc = fmaf(buf[k + 1], w1, acc);
        acc = fmaf(buf[k + 2], w2, acc);
        acc = fmaf(buf[k + 3], w3, acc);
      }
      for (; k < b; ++k) acc = fmaf(buf[k], w[k], acc);
      return acc;
    };
    if (2 * nmel_pad <= L) {
      // fewer filters than half the frame's lanes: L / nmel_pad lanes share a filter, each a stretch of its band
      // (the bands of a long transform are long: 26 filters on 1025 bins reach 190 bins), partial sums meet by shuffles
      const int j = li & (nmel_pad - 1), part = li / nmel_pad, parts = L / nmel_pad;
      float acc = 0.f;
      if (j < nmel) {
        const int lo = bandl[3 * j], len = bandl[3 * j + 1] - lo;
        acc = band_sum(j, lo + len * part / parts, lo + len * (part + 1) / parts);
      }
      for (int m = nmel_pad; m < L; m <<= 1) acc += __shfl_xor(acc, m, L);
      if (j < nmel && part == 0) {
        if (gc.floor_mel && acc == 0.f) acc = kEps64;
        lrow[j] = fast_ln(acc);
      }
    } else {
      for (int j = li; j < nmel; j += L) {
        float acc = band_sum(j, bandl[3 * j], bandl[3 * j + 1]);
        if (gc.floor_mel && acc == 0.f) acc = kEps64;
        lrow[j] = fast_ln(acc);
      }
    }
    wave_sync();
    const bool own = live && t >= t0 && t < t0 + nt;
    if (out_kind == kOutLogmel) {
      if (own) {
        float* const o = p.out + (f_off + t) * ld;
        for (int j = li; j < nmel; j += L) o[j] = lrow[j];
        if (li == 0) o[nmel] = lne;
      }
      wave_sync();
      continue;
    }
    // ---- DCT-II * lifter, c0 := ln E (mfcc-self5.py:136-167)
    for (int c = li; c < ncep; c += L) {
      float acc = 0.f;
      int n = 0;
      for (; n + 4 <= nmel; n += 4) {
        const float w0 = dctw[n * ncep + c], w1 = dctw[(n + 1) * ncep + c], w2 = dctw[(n + 2) * ncep + c], w3 = dctw[(n + 3) * ncep + c];
        acc = fmaf(lrow[n], w0, acc);
        acc = fmaf(lrow[n + 1], w1, acc);
        acc = fmaf(lrow[n + 2], w2, acc);
        acc = fmaf(lrow[n + 3], w3, acc);
      }
      for (; n < nmel; ++n) acc = fmaf(lrow[n], dctw[n * ncep + c], acc);
      if (c == 0 && !gc.keep_c0) acc = lne;
      if (full) {
        if (live) stat[(t - s_lo) * ncep + c] = acc;
      } else if (own) {
        p.out[(f_off + t) * ld + c] = acc;
      }
    }
    wave_sync();
  }
  if (!full) return;
  general_tile_rows(gc, stat, dlt, s_lo, t0, nt, nframes, p.out + f_off * ld, ld, tid);
}

}  // namespace mfcc
