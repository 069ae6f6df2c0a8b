// MFCC-39 kernel with the work of a workgroup split over two sets of waves (gfx950).
//
// fused_kernel.h gives every wave all of a batch's work in turn: transform, filterbank, rows.  Its 168
// registers and 52 KB of LDS allow three waves per SIMD, and at three waves per SIMD a vector instruction of a
// mixed stream costs about six cycles of issue; at four it costs ten per cent less
// (tools/ubench/occupancy_rate.hip).  Here a workgroup has EIGHT waves in two roles, so that neither carries
// the other's registers, and two workgroups share a CU (four waves per SIMD: one of each role per workgroup):
//
//   waves 0-3  "T"   transform batch k: frames from the staged audio, the two 16-point passes with the
//                    exchange through LDS, untangle, spectrum rows and energy sums into LDS;
//   waves 4-7  "F"   filterbank, logarithm and chained DCT of batch k - 1 (matrix instructions, as in
//                    fused_kernel.h), the output rows whose context is complete, staging of the audio of
//                    batch k + 1, sum of the DCT shares into the ring of static rows.
//
// A step of the pipeline has two workgroup barriers:
//
//        T                                        F
//   frames of batch k out of A, pass 1,      loads of A(k+1) issued; operands and energies of S(k-1) read;
//   exchange, pass 2                         filterbank + DCT -> shares; rows out
//   ---------------------------------------- X ----------------------------------------------------------
//   untangle -> S(k), energy sums            A(k+1) converted and written; shares summed -> ring
//   ---------------------------------------- Y ----------------------------------------------------------
//
// A = staged audio (one buffer: read before X, rewritten after), S = spectrum rows (one buffer: read before X,
// rewritten after; the shares and energies sit beside the rows in parts of a slot the untangle does not touch).
//
// A workgroup is PERSISTENT: it walks a list of tiles (FusedParams::wg_tiles, cut on the host so that every
// workgroup of the grid gets the same number of batches) and the pipeline runs on across tile boundaries -- T
// transforms the first batch of the next tile while F finishes the last batch of this one -- so the tables are
// loaded once per workgroup, no tile waits for its first samples, and the grid ends level.
// The tile list, the tables, the staging, the transform, the filterbank and the row output are those of
// fused_kernel.h; results are bit-identical to its MFCC-39 kind.
#pragma once

#include "fused_kernel.h"

namespace mfcc {

// Spectrum slot of a frame: the row (bins behind a skew of 0..48 floats, as in fused_kernel.h), the four
// waves' DCT shares at kSPartBase.., the frame's energy sum at kSPartEnergy.  384 floats = 6 x 64 banks, so the
// bank pattern of the rows is that of the 512-float slots of fused_kernel.h.
constexpr int kSSlot = 384;
constexpr int kSPartBase = 320, kSPartEnergy = 316;
static_assert(48 + kRowPad <= kSPartEnergy, "row, energy and shares of a slot do not overlap");
__host__ __device__ constexpr int spart_index(int slot, int w, int c) {
  return slot * kSSlot + kSPartBase + ((16 * w + (c & ~3) + 16 * (slot & 1) + 8 * ((slot >> 1) & 1)) & 63) + (c & 3);
}
constexpr int kSbufFloats = kBatch * kSSlot;
// Exchange buffer of a frame between the two 16-point passes: 16 rows (first-pass output index k1) of 16 complex
// values, rows 136 bytes apart, frames 2176 bytes.  A lane writes column q of every row and reads row q: with rows
// 34 banks apart the sixteen 8-byte reads of a frame's lanes fall on 32 different banks, the next frame's (32
// banks further) on the other 32, and a write instruction covers 128 contiguous bytes per frame -- no conflicts
// either way, and every address is one per-lane base plus a constant (fused_kernel.h keeps sixteen xor-swizzled
// addresses in registers instead; this kernel has 128 registers, not 168).
constexpr int kERow = 136, kEFrame = 16 * kERow;
constexpr int kEbufFloats = kBatch * kEFrame / 4;
constexpr int split_smem_floats(int ring_pitch) {
  return kEbufFloats + kSbufFloats + kAbufFloats + ring_rows(ring_pitch) * ring_pitch + kTwFloats + kDctFloats;
}
static_assert(split_smem_floats(13) * 4 <= 64 * 1280 && split_smem_floats(kMaxCep) * 4 <= 64 * 1280, "two workgroups per CU");

// What a role needs to know about a tile (all wave-uniform).
struct SplitTile {
  int64_t s_off, len, f_off;
  int nframes, t0, nt, s_lo, s_hi, nb;
};
__device__ __forceinline__ SplitTile split_tile(const FusedParams& p, int idx) {
  const Tile tile = p.wg_tiles[idx];
  SplitTile c;
  c.s_off = p.sample_off[tile.utt];
  c.len = p.sample_len ? p.sample_len[tile.utt] : p.sample_off[tile.utt + 1] - c.s_off;
  c.f_off = p.frame_off[tile.utt];
  c.nframes = (int)(p.frame_off[tile.utt + 1] - c.f_off);
  c.t0 = tile.t0;
  c.nt = tile.nt;
  c.s_lo = max(0, c.t0 - kHalo);
  c.s_hi = min(c.nframes, c.t0 + c.nt + kHalo);
  c.nb = (c.s_hi - c.s_lo + kBatch - 1) / kBatch;
  return c;
}

#ifdef MFCC_DIAG
#undef MFCC_STAMP
#define MFCC_STAMP(var) const unsigned long long var = __builtin_readcyclecounter()
#endif

template <int NCEP, typename T, class G>
__global__ void __launch_bounds__(512, 4) split_kernel(const FusedParams p) {
  extern __shared__ __attribute__((aligned(128))) float smem[];
  float* const ebuf = smem;                       // exchange buffers of the transforms
  float* const sbuf = ebuf + kEbufFloats;         // spectrum slots
  float* const abuf = sbuf + kSbufFloats;         // staged audio
  constexpr int RP = NCEP > 0 ? NCEP : kMaxCep;
  constexpr bool kMirror = ring_rows(RP) > 32;
  float* const ring = abuf + kAbufFloats;
  float* const twl = ring + ring_rows(RP) * RP;
  float* const dctl = twl + kTwFloats;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool role_t = wave8 < 4;
  const int wave = wave8 & 3;
  const int rtid = tid & 255;  // thread index within the role
  const int q = lane & 15;
  const int g4 = lane >> 4;
  const int fb = wave * 4 + g4;

  const PlanConst* const pc = p.pc;
  const int tile_first = p.wg_off[blockIdx.x], tile_end = p.wg_off[blockIdx.x + 1];
  if (tile_first >= tile_end) return;  // (all waves of the workgroup alike)
  const int ncep = NCEP > 0 ? NCEP : pc->numcep;
  const int ld = (int)p.out_ld;
  const float pre_hi = pc->pre_hi, pre_lo = pc->pre_lo;
  const bool mag = pc->spectrum != 0;
  const float spec_scale = pc->spec_scale;
  const bool half_band = pc->decim == 2;
  const gfloat_ptr window = as_global(pc->window);
  const bool floor_mel = pc->floor_mel != 0;

  {
    const gfloat_ptr tw = as_global(pc->tw512);
    for (int i = tid; i < 16 * 23; i += 512) {
      const int qq = i / 23, k = i - qq * 23;
      const int m = k < 15 ? (2 * qq * (k + 1)) & 511 : 512 + qq + 16 * (k - 15);  // untangle entries: (tan, cos), behind the 512 powers
      const int slot = k < 15 ? kTwSt1 + k : kTwUnt + (k - 15);
      *(cf*)(twl + qq * kTwPitch + 2 * slot) = mk(tw[2 * m], tw[2 * m + 1]);
    }
    const gfloat_ptr dwg = as_global(pc->dctW);
    for (int i = tid; i < kDctFloats; i += 512) dctl[i] = dwg[i];
  }
  const cf* const twq = (const cf*)(twl + q * kTwPitch);
  const int partner_addr = ((lane & 48) | partner_lane(q)) << 2;

#ifdef MFCC_DIAG
  unsigned long long dsum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  MFCC_STAMP(tk0);
#define MFCC_STEP_STATS()     \
  do {                        \
    MFCC_STAMP(ts4);          \
    dsum[0] += ts1 - ts0;     \
    dsum[1] += ts2 - ts1;     \
    dsum[2] += ts3 - ts2;     \
    dsum[3] += ts4 - ts3;     \
    dsum[4] += 1;             \
  } while (0)
#else
#define MFCC_STEP_STATS()
#endif

  // One loop per role, each with the step's two barriers (a barrier counts the waves that arrive, wherever in the
  // program they do): the registers a role keeps across its loop -- the transform's values and addresses there,
  // the filter weights and the row writer's addresses here -- are then allocated side by side, not on top of
  // each other.  Both loops run B + 1 steps, B = the batches of the workgroup's tiles: T is idle in the last
  // step, F transforms nothing in the first.
  if (role_t) {
    __builtin_amdgcn_s_setprio(MFCC_PRIO_FFT);
    int ti = tile_first, k = 0;
    SplitTile c = split_tile(p, ti);
    // alignment offset of the tile's staged audio, as stage_setup derives it
    auto delta_of = [&](const SplitTile& t) {
      return (int)(((uintptr_t)((const T*)p.audio + t.s_off + (int64_t)t.s_lo * G::kStep) & 15) / sizeof(T));
    };
    int delta = delta_of(c);
    __syncthreads();  // tables and the first batch's audio are in LDS
    for (;;) {
      MFCC_STAMP(ts0);
      const bool t_on = ti < tile_end;
      cf x[16];
      if (t_on) {
        const float* fa = abuf + delta + fb * G::kStep;
        if (delta & 1) {
#pragma unroll
          for (int n1 = 0; n1 < 16; ++n1) {
            const int n = 16 * n1 + q;
            if (2 * (16 * n1) < G::kSize) {
              const int nn = min(n, G::kSize / 2 - 1);
              const float ok = 2 * n + 1 < G::kSize ? 1.f : 0.f;
              x[n1] = mk(fa[2 * nn] * ok, fa[2 * nn + 1] * ok);
            } else {
              x[n1] = mk(0.f, 0.f);
            }
          }
        } else {
          const cf* fa2 = (const cf*)__builtin_assume_aligned(fa, 8);
          constexpr int kFull = (G::kSize / 2) / 16;
          const unsigned xa = lds_addr(fa2 + q);
#pragma unroll
          for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int a = 0; a < 4; ++a) {
              const int n1 = 4 * a + b;
              if (n1 < kFull) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(x[n1]) : "v"(xa), "i"(n1 * 128));
            }
#pragma unroll
          for (int n1 = kFull; n1 < 16; ++n1) {
            const int n = 16 * n1 + q;
            if (2 * (16 * n1) < G::kSize) {
              const float ok = 2 * n + 1 < G::kSize ? 1.f : 0.f;
              x[n1] = fa2[min(n, G::kSize / 2 - 1)] * mk(ok, ok);
            } else {
              x[n1] = mk(0.f, 0.f);
            }
          }
          static_assert(kFull == 12 || kFull == 6, "wait counts below are written for 400- and 200-sample frames");
          if (kFull == 12) {
            asm volatile("s_waitcnt lgkmcnt(9)" : "+v"(x[0]), "+v"(x[4]), "+v"(x[8]));
            asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(x[1]), "+v"(x[5]), "+v"(x[9]));
            asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(x[2]), "+v"(x[6]), "+v"(x[10]));
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(x[3]), "+v"(x[7]), "+v"(x[11]));
          } else {
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]));
          }
        }
        if (window) {
          cf wv[(G::kSize + 31) / 32];
#pragma unroll
          for (int n1 = 0; n1 < (G::kSize + 31) / 32; ++n1) wv[n1] = mk(window[2 * (16 * n1 + q)], window[2 * (16 * n1 + q) + 1]);
#pragma unroll
          for (int n1 = 0; n1 < (G::kSize + 31) / 32; ++n1) x[n1] = x[n1] * wv[n1];
        }
        cf tw1[16];
        {
          const float4* const tv = (const float4*)__builtin_assume_aligned(twq + kTwSt1, 16);
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float4 v = tv[j];
            tw1[2 * j] = mk(v.x, v.y);
            tw1[2 * j + 1] = mk(v.z, v.w);
          }
        }
        fft16_pruned<(G::kSize + 31) / 32>(x);
        char* const lds0 = (char*)ebuf;
        char* const wp = lds0 + fb * kEFrame + (q << 3);
        const char* const rp = lds0 + fb * kEFrame + q * kERow;
#pragma unroll
        for (int k1 = 0; k1 < 16; ++k1) {
          if (k1) x[k1] = cmul(x[k1], tw1[k1 - 1]);
          *(cf*)(wp + k1 * kERow) = x[k1];
        }
        wave_sync();
#pragma unroll
        for (int n2 = 0; n2 < 16; ++n2) x[n2] = *(const cf*)(rp + 8 * n2);
        fft16(x);
      }
      MFCC_STAMP(ts1);
      __syncthreads();  // X
      MFCC_STAMP(ts2);
      if (t_on) {
        float* const P = sbuf + fb * kSSlot + row_skew(fb);
        cf esum = mk(0.f, 0.f);
#pragma unroll
        for (int h = 0; h < 8; h += 4) {
          cf pb[4], pw[4];
#pragma unroll
          for (int s = 0; s < 4; ++s) pb[s] = mk(lane_read(partner_addr, x[15 - h - s].x), lane_read(partner_addr, x[15 - h - s].y));
          {
            const float4* const tv = (const float4*)__builtin_assume_aligned(twq + kTwUnt + h, 16);
            const float4 v0 = tv[0], v1 = tv[1];
            pw[0] = mk(v0.x, v0.y); pw[1] = mk(v0.z, v0.w); pw[2] = mk(v1.x, v1.y); pw[3] = mk(v1.z, v1.w);
          }
          float lo4[4], hi4[4];
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            if (q == 0) pb[s] = x[(16 - h - s) & 15];
            untangle_power_tc(x[h + s], pb[s], pw[s], lo4[s], hi4[s]);
          }
          if (mag) {
            asm volatile("" ::: "memory");
#pragma unroll
            for (int s = 0; s < 4; ++s) {
              lo4[s] = __builtin_amdgcn_sqrtf(lo4[s]);
              hi4[s] = __builtin_amdgcn_sqrtf(hi4[s]);
            }
          }
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            P[q + 16 * (h + s)] = lo4[s];
            P[256 - q - 16 * (h + s)] = hi4[s];
            esum = esum + mk(lo4[s], hi4[s]);
          }
        }
        if (q == 0) {
          const float p128 = 4.f * norm2(x[8]);
          const float v128 = mag ? __builtin_amdgcn_sqrtf(p128) : p128;
          P[128] = v128;
          P[257] = 0.f; P[258] = 0.f; P[259] = 0.f;
          esum.x += v128;
        }
        float e = esum.x + esum.y;
        if (half_band && (q & 1)) e = 0.f;
        e += row_ror<8>(e);
        e += row_ror<4>(e);
        e += row_ror<2>(e);
        e += row_ror<1>(e);
        if (q == 0) sbuf[fb * kSSlot + kSPartEnergy] = e;
      }
      MFCC_STAMP(ts3);
      __syncthreads();  // Y
      MFCC_STEP_STATS();
      if (!t_on) break;  // that was the step in which F finished the last batch
      if (++k == c.nb) {
        k = 0;
        if (++ti < tile_end) {
          c = split_tile(p, ti);
          delta = delta_of(c);
        }
      }
    }
  } else {
    // ---- F: two tiles in view -- the one whose audio is being staged (a batch ahead of T) and the one whose
    //      spectrum rows are being turned into feature rows (a batch behind T)
    const int mel_nslots = pc->map.nslots[wave];
    unsigned slot_info[kMelQuads];
#pragma unroll
    for (int s = 0; s < kMelQuads; ++s) slot_info[s] = pc->map.slot[wave][s];
    typedef const f32x4 __attribute__((address_space(1))) * gfloat4_ptr;
    const gfloat4_ptr melW4 = (gfloat4_ptr)(uintptr_t)pc->melW + (wave * kMelQuads) * 16 + (lane & 15);
    f32x4 bv[kMelQuads];  // the filter weights of this wave's slots (A operands), read once per workgroup
#pragma unroll
    for (int s = 0; s < kMelQuads; ++s) bv[s] = melW4[s * 16];
    const int frm = (lane >> 4) * 4 + (lane & 3);
    const int blk = (lane >> 2) & 3;
    const float* const Prow = sbuf + frm * kSSlot + row_skew(frm) + 4 * blk;
    const float* const dct_lane = dctl + 4 * (lane & 15);

    // staging side
    int tis = tile_first, ks = 0;
    SplitTile cs = split_tile(p, tis);
    StageRegs sr;
    StageCtx<T> sctx = stage_setup<T, G>((const T*)p.audio, p.total_elems, cs.s_off, cs.s_lo);
    stage_load<T, G>(sctx, sr, 0, rtid);
    stage_store<T, G>(sr, abuf, cs.len, cs.s_lo, rtid, pre_hi, pre_lo);
    auto next_stage = [&]() {  // the batch after the one just staged; false when there is none
      if (++ks == cs.nb) {
        ks = 0;
        if (++tis >= tile_end) return false;
        cs = split_tile(p, tis);
        sctx = stage_setup<T, G>((const T*)p.audio, p.total_elems, cs.s_off, cs.s_lo);
      }
      return true;
    };
    bool stage_next = next_stage();
    // feature side
    int tim = tile_first, km = 0;
    SplitTile cm = split_tile(p, tim);
    float* out_t0 = p.out + (cm.f_off + cm.t0) * p.out_ld;
    __syncthreads();  // tables and the first batch's audio are in LDS

    auto regress = [](float a, float b, float m, float d, float e) { return regress5(a, b, m, d, e); };
    // rows r_first .. r_first + 15 of the feature side's utterance (those inside its tile): as emit_rows of
    // fused_kernel.h
    auto emit_rows = [&](int r_first) {
      const int nframes = cm.nframes, s_lo = cm.s_lo, t0 = cm.t0, nt = cm.nt;
      const int r = r_first + 4 * wave + g4;
      const int r_wave = r_first + 4 * wave;
      const bool interior = r_wave >= kHalo && r_wave + 3 + kHalo <= nframes - 1;
      float v[9], d[5];
      if (interior) {
        const int b0 = r - kHalo - s_lo + 8;
        const float* const w0 = ring + (b0 & 31) * RP + q;
        auto at = [&](int m) { return kMirror ? w0[m * RP] : ring[((b0 + m) & 31) * RP + q]; };
        cf pe[4], po[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          pe[m] = mk(at(2 * m), at(2 * m + 1));
          po[m] = mk(at(2 * m + 1), at(2 * m + 2));
        }
        const cf two = mk(2.f, 2.f), zero = mk(0.f, 0.f), tenth = mk(0.1f, 0.1f);
        auto regress2 = [&](cf a, cf b, cf m, cf dd, cf e) {
          return __builtin_elementwise_fma(zero, m, __builtin_elementwise_fma(two, e - a, dd - b)) * tenth;
        };
        const cf d01 = regress2(pe[0], po[0], pe[1], po[1], pe[2]);
        const cf d23 = regress2(pe[1], po[1], pe[2], po[2], pe[3]);
        v[4] = pe[2].x;
        d[0] = d01.x; d[1] = d01.y; d[2] = d23.x; d[3] = d23.y;
        d[4] = regress(pe[2].x, pe[2].y, pe[3].x, pe[3].y, po[3].y);
      } else {
        asm volatile("" ::: "memory");
#pragma unroll
        for (int m = 0; m < 9; ++m) {
          const int t = min(max(r - kHalo + m, 0), nframes - 1);
          v[m] = ring[((t - s_lo + 8) & 31) * RP + q];
        }
        float e[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) e[k] = regress(v[k], v[k + 1], v[k + 2], v[k + 3], v[k + 4]);
        const int u = nframes - 1 - r;
        d[2] = e[2];
        d[1] = r == 0 ? e[2] : e[1];
        d[0] = r == 0 ? e[2] : (r == 1 ? e[1] : e[0]);
        d[3] = u == 0 ? e[2] : e[3];
        d[4] = u == 0 ? e[2] : (u == 1 ? e[3] : e[4]);
      }
      if (q < ncep && r >= t0 && r < t0 + nt) {
        float* const o = out_t0 + (r - t0) * ld + q;
        o[0] = v[4];
        o[ncep] = d[2];
        o[2 * ncep] = regress(d[0], d[1], d[2], d[3], d[4]);
      }
    };

    for (int step = 0;; ++step) {
      MFCC_STAMP(ts0);
      const bool f_on = step >= 1;            // the feature side runs a step behind T
      const int tbf = cm.s_lo + km * kBatch;  // its batch
      float lne = 0.f;
      // (issuing these loads a whole step earlier, right behind the store of the batch before, measured 2 % slower)
      if (stage_next) stage_load<T, G>(sctx, sr, ks * kBatch * G::kStep, rtid);
      __builtin_amdgcn_sched_barrier(0);
      if (f_on) {
        __builtin_amdgcn_s_setprio(MFCC_PRIO_MEL);
        // the energy of the frame this lane finishes after X (T overwrites the slot's sum there)
        {
          float e = sbuf[(4 * wave + g4) * kSSlot + kSPartEnergy] * spec_scale;
          if (e == 0.f) e = kEps64;  // mfcc-self5.py:130-132
          lne = fast_ln(e);
        }
        f32x4 pd = {0.f, 0.f, 0.f, 0.f};
        f32x4 pa[kMelQuads];
#pragma unroll
        for (int s = 0; s < kMelQuads; ++s) {
          const float4 v = *(const float4*)__builtin_assume_aligned(Prow + 4 * (slot_info[s] & 63), 16);
          pa[s] = f32x4{v.x, v.y, v.z, v.w};
        }
        f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f}, dwc = {0.f, 0.f, 0.f, 0.f};
        int nslots_now = __builtin_amdgcn_readfirstlane(mel_nslots);
        asm volatile("" : "+s"(nslots_now));
#pragma unroll
        for (int s = 0; s < kMelQuads; ++s) {
          if (s >= nslots_now) break;
          unsigned info = __builtin_amdgcn_readfirstlane(slot_info[s]);
          asm volatile("" : "+s"(info));
          if (info & kSlotFirst) {
            acc = f32x4{0.f, 0.f, 0.f, 0.f};
            acc2 = f32x4{0.f, 0.f, 0.f, 0.f};
            dwc = *(const f32x4*)__builtin_assume_aligned(dct_lane + 64 * ((info >> 17) & 15), 16);
          }
          acc = __builtin_amdgcn_mfma_f32_4x4x1f32(bv[s][0], pa[s][0], acc, 0, 0, 0);
          acc2 = __builtin_amdgcn_mfma_f32_4x4x1f32(bv[s][1], pa[s][1], acc2, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_4x4x1f32(bv[s][2], pa[s][2], acc, 0, 0, 0);
          acc2 = __builtin_amdgcn_mfma_f32_4x4x1f32(bv[s][3], pa[s][3], acc2, 0, 0, 0);
          if (info & kSlotLast) {
            f32x4 mv = acc + acc2;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              mv[r] += row_ror<4>(mv[r]);
              mv[r] += row_ror<8>(mv[r]);
            }
            const int nf = (info >> 8) & 7;
            if (floor_mel) {
              asm volatile("" ::: "memory");
#pragma unroll
              for (int r = 0; r < 4; ++r) mv[r] = mv[r] == 0.f ? kEps64 : mv[r];
            }
            float lm[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) lm[r] = __builtin_amdgcn_logf(mv[r]);
            if (nf < 4) {
              asm volatile("" ::: "memory");
#pragma unroll
              for (int r = 1; r < 4; ++r) lm[r] = r < nf ? lm[r] : 0.f;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) pd = __builtin_amdgcn_mfma_f32_4x4x1f32(dwc[r], lm[r], pd, 0, 0, 0);
          }
        }
        *(f32x4*)__builtin_assume_aligned(sbuf + spart_index(frm, wave, 4 * blk), 16) = pd;
#ifdef MFCC_DIAG
        {
          MFCC_STAMP(tmel);
          dsum[5] += tmel - ts0;  // staging loads issued, operands read, filterbank + DCT
        }
#endif
        // rows of the batch before this one whose +-4-row context is complete
        if (km > 0) emit_rows(tbf - kBatch - kHalo);
      }
      MFCC_STAMP(ts1);
      __syncthreads();  // X
      MFCC_STAMP(ts2);
      __builtin_amdgcn_s_setprio(MFCC_PRIO_P3);
      if (stage_next) {
        stage_store<T, G>(sr, abuf, cs.len, cs.s_lo + ks * kBatch, rtid, pre_hi, pre_lo);
        stage_next = next_stage();
      }
#ifdef MFCC_DIAG
      {
        MFCC_STAMP(tst);
        dsum[6] += tst - ts2;  // staged audio converted and written
      }
#endif
      if (f_on) {
        // sum of the four DCT shares, c0 := ln E; wave w finishes frames 4 w .. 4 w + 3 of its batch
        const int i = 4 * wave + g4;
        const int t = tbf + i;
        float v = (sbuf[spart_index(i, 0, q)] + sbuf[spart_index(i, 1, q)]) + (sbuf[spart_index(i, 2, q)] + sbuf[spart_index(i, 3, q)]);
        if (q == 0) v = lne;
        if (q < ncep && t < cm.s_hi) {
          const int rho = (t - cm.s_lo + 8) & 31;
          ring[rho * RP + q] = v;
          if (kMirror && rho < 8) ring[(rho + 32) * RP + q] = v;
        }
      }
      MFCC_STAMP(ts3);
      __syncthreads();  // Y
      MFCC_STEP_STATS();
      if (f_on && ++km == cm.nb) {
        // the tile's last batch is in the ring: the rows still held back, from 4 rows before that batch to the
        // tile's end (at most 20), leave now -- the next tile's rows enter the ring after the next X
        const int tb_last = cm.s_lo + (cm.nb - 1) * kBatch;
        emit_rows(tb_last - kHalo);
        if (tb_last + kBatch - kHalo < cm.t0 + cm.nt) emit_rows(tb_last + kBatch - kHalo);
        km = 0;
        if (++tim >= tile_end) break;
        cm = split_tile(p, tim);
        out_t0 = p.out + (cm.f_off + cm.t0) * p.out_ld;
      }
    }
  }
#undef MFCC_STEP_STATS
#ifdef MFCC_DIAG
  {
    MFCC_STAMP(tk2);
    dsum[7] = tk2 - tk0;
    if (lane == 0 && p.dbg)
      for (int i = 0; i < 8; ++i) p.dbg[((size_t)blockIdx.x * 8 + wave8) * 8 + i] = dsum[i];
  }
#endif
}

}  // namespace mfcc
