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
//                  filterbank of batch b runs;
//               2. each wave transforms 4 frames, 16 lanes per frame (fft_lane.h), and leaves
//                  the 257 spectrum values of each frame in LDS;                [barrier]
//               3. the [filters x 257] x [257 x 16 frames] filterbank product runs as matrix
//                  instructions on 4x4 blocks (v_mfma_f32_4x4x1_16b_f32, an exact f32 fma chain):
//                  the filters are cut into groups of four, whose weights cover a short range of
//                  bins (the bank is a band matrix); every wave owns some of the groups, takes the
//                  logarithm of its sums at once and feeds them, still in registers, to its share
//                  of the [numcep x filters] DCT*lifter product; the four shares go to a small
//                  LDS tile;                                                      [barrier]
//               4. the shares are summed, c0 := ln E, and the static row enters a ring of 32
//                  rows in LDS.
//   rows out  one batch later, when the rows t-4 .. t+4 of a frame exist, delta and delta-delta
//             are formed from the ring and the whole row c | delta | delta-delta is stored: every
//             output line is written once, nothing is read back (4 halo frames are computed on
//             each side of a tile where the utterance continues).
//
// On gfx950 the fp32 matrix instructions run on the vector ALU's pipe (a wave mix of both takes
// the sum of their times) at 32 multiply-adds per cycle whatever their shape, so what counts is
// how many of an instruction's multiply-adds meet a non-zero weight.  A bin feeds two filters
// (mfcc-self5.py:102-117): in a 4-filter block half of the products are useful, in the
// 16-filter block of v_mfma_f32_16x16x4_f32 an eighth.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "fft_lane.h"

// Issue priority of a wave per phase of the batch loop (s_setprio): the short filterbank / DCT
// phase ahead of the long FFT phase of the other workgroups' waves on the same SIMD, so that a
// workgroup reaches its second barrier early; measured 1.6 % over equal priorities.
#ifndef MFCC_PRIO_FFT
#define MFCC_PRIO_FFT 1
#endif
#ifndef MFCC_PRIO_MEL
#define MFCC_PRIO_MEL 3
#endif
#ifndef MFCC_PRIO_P3
#define MFCC_PRIO_P3 0
#endif

#ifndef MFCC_MIN_WAVES
#define MFCC_MIN_WAVES 3  // waves per SIMD the register allocation must allow (3 workgroups per CU)
#endif

namespace mfcc {

// Frame geometry is a compile-time parameter of the kernels.  400/160 is the reference's
// int(16000 * 0.025) / int(16000 * 0.010) (mfcc-self5.py:75-76); 200/80 is the same 25 ms / 10 ms
// at 8 kHz with the FFT kept at 512 points, as the reference's CTC scripts ask of their MFCC
// routine (inference_ctc_eri_05.py:96,253-257: RATE = 8000, nfft=512).
template <int SIZE, int STEP>
struct FrameGeom {
  static constexpr int kSize = SIZE;
  static constexpr int kStep = STEP;
  static constexpr int kStageLen = 15 * STEP + SIZE;  // samples staged per 16-frame batch
};
typedef FrameGeom<400, 160> Geom16k;
typedef FrameGeom<200, 80> Geom8k;
// 20 ms frames every 10 ms at 16 kHz (frame_time = 20: int(16000 * 0.020) = 320 samples, mfcc-self5.py:75), FFT 512.
// Instantiated for the kinds of mfcc_forward only (not for the stacked and streaming kinds).
typedef FrameGeom<320, 160> Geom20ms;
constexpr int kFrameSize = Geom16k::kSize;   // largest supported frame: sizes the LDS carve
constexpr int kFrameStep = Geom16k::kStep;
constexpr int kNfft = 512;        // next power of two, mfcc-self5.py:79-84
constexpr int kNbin = kNfft / 2 + 1;
constexpr int kBatch = 16;        // frames per batch (4 waves x 4 frames)
constexpr int kHalo = 4;          // delta-delta reach: 2 * pad_fac
constexpr int kMaxCep = 16;
constexpr int kMaxMel = 40;       // "26-40 Mel bins" (mfcc-self5.py:23); bounds the LDS tables
constexpr int kStage = (kBatch - 1) * kFrameStep + kFrameSize;  // 2800 samples per batch
constexpr int kMelGroups = kMaxMel / 4;  // filter groups (up to four filters each: the rows of a 4x4 block) at most
constexpr int kMelQuads = 6;      // slots of four filterbank k-steps (16 bins) a wave runs at most

// LDS carve (floats).
constexpr int kSlot = 512;                           // LDS floats per frame slot: the 2 KiB exchange buffer of a frame
// Spectrum row of frame slot r starts at r*512 + row_skew(r) floats.  The skew (a multiple of 4
// floats, so rows stay 16-byte aligned) serves both access patterns of the rows without bank
// conflicts: the untangle writes 16 consecutive columns from each of a wave's 4 frames with
// ds_write_b32 (rows 4w and 4w+1, resp. 4w+2 and 4w+3, share an instruction half: their skews
// differ by 16 banks); the filterbank reads 16 bytes per lane with ds_read_b128, lane
// 16 a + 4 b + c taking row 4 a + c at 16-byte column block b: the instruction's 16-lane groups
// {0-3, 12-15, 20-27} and {4-11, 16-19, 28-31} then hold blocks {0, 3} of rows 0-3 with blocks
// {1, 2} of rows 4-7, resp. the other way round -- with the rows of a group of four 64 bytes
// apart, each lane group covers all sixteen 16-byte slots of the 64 banks once.
__host__ __device__ constexpr int row_skew(int r) { return 16 * (r & 3); }
constexpr int kRowPad = 260;                         // bins 0..256 plus zeros up to a multiple of 4
constexpr int kXbufFloats = kBatch * kSlot;          // exchange buffers (fb*512) / spectrum rows
constexpr int kAbufFloats = 2808;                    // staged audio: kStage + up to 7 alignment samples, whole chunks
// Partial DCT sums (one share per wave, the filters it owns) of frame slot i live in the upper part
// of that slot's 2 KiB (the spectrum row ends below float 320, the exchange of the next transform
// is written by the very wave that reads the shares, later in its program): coefficient c of wave w
// at float kPartBase + ((16 w + (c & ~3) + rot(i)) mod 64) + (c & 3), rot = 0, 16, 8, 24 for i mod 4 =
// 0..3 -- a lane's four coefficients are one 16-byte write, the eight lanes of a write group (two
// coefficient blocks of four frames) land on 32 different banks, and the two frames whose sixteen
// coefficients one half of a ds_read_b32 collects lie 16 banks apart -- and the frame's energy sum
// (formed beside the transform) at kPartEnergy.
constexpr int kPartBase = 320, kPartEnergy = 384;
static_assert(48 + kRowPad <= kPartBase, "the shares of a frame slot start behind its spectrum row");
__host__ __device__ constexpr int part_index(int slot, int w, int c) {
  return slot * kSlot + kPartBase + ((16 * w + (c & ~3) + 16 * (slot & 1) + 8 * ((slot >> 1) & 1)) & 63) + (c & 3);
}
// Ring of static-feature rows for the delta / delta-delta regression: row t of the tile sits at
// ring row (t - s_lo + 8) mod 32; ring rows 0..7 are mirrored at 32..39 so that the 9 rows
// t-4 .. t+4 a lane reads are always consecutive.
constexpr int kRingRows = 40;
// Kernels with a run-time numcep keep kMaxCep floats per ring row; 32 rows of those fill the same LDS
// as 40 rows of 13, so they do without the mirror rows and wrap every row index instead.
constexpr int ring_rows(int ring_pitch) { return ring_pitch > 13 ? 32 : kRingRows; }
// per-lane-index twiddle row: 8 untangle values, then 15 stage-1 values, read 16 bytes at a time;
// 52 q mod 64 are 16 distinct multiples of 4, so the 16 rows a read touches cover all 64 banks
constexpr int kTwPitch = 52;
constexpr int kTwUnt = 0, kTwSt1 = 8;                // complex index of the first untangle / stage-1 value
constexpr int kTwFloats = 16 * kTwPitch;
// DCT * lifter weights of the chained product, A operands of v_mfma_f32_4x4x1_16b_f32: group g at
// float 64 g, 16 bytes per lane index (lane & 15)
constexpr int kDctFloats = kMelGroups * 64;
// Stacked MFCC-39 kind: ring of finished rows (3 numcep floats each, back to back): row r of the tile's rows at
// ring row (r - first) mod 64, rows 0 .. kStackMaxN - 2 repeated behind row 63 so that the N rows of a context
// window are always one contiguous run of floats.
constexpr int kORows = 64;
constexpr int oring_floats(int ncep) { return (kORows + 20) * 3 * ncep; }
// kOutMfcc39Sums: per wave, sums of c - K, delta-delta, delta and of the squares of c - K, delta, delta-delta,
// 16 columns each (column q of block k at 16 k + q: the order in which the lanes' reduction leaves them); the
// four waves' arrays are added in a fixed order when the tile is done.  K = the
// static row of the tile's first frame (0 where that is not finite): the cepstra, c0 = ln E above all, have a
// mean far from zero, and float32 sums of x^2 would lose the variance in it; delta rows have none.  A tile
// leaves its 6 x 16 sums and the 16 values of K (kTileSums floats).
constexpr int kTileSumCols = 6 * 16;
constexpr int kTileSums = kTileSumCols + 16;
static_assert(kTileSumCols == 6 * kMaxCep, "one column per coefficient a plan may have");
constexpr int kSumsLdsFloats = 4 * kTileSumCols + 16;
// ring_pitch = floats per ring row: numcep where it is a compile-time constant, else kMaxCep
constexpr int smem_floats(int ring_pitch) { return kXbufFloats + kAbufFloats + ring_rows(ring_pitch) * ring_pitch + kTwFloats + kDctFloats; }
// Stacked spectrum kinds: instead of the partial-DCT tiles and the statics ring, a ring of the last
// spectrum rows (frame s of the tile at ring row (s - s_lo) mod rows, 257 values behind a phase of
// 0..3 floats), from which whole output rows are written.  The ring holds the 2 * (N / 2) frames a row
// reaches back plus the batch being transformed: 32 rows for windows of up to 17 frames (81 632 bytes:
// two workgroups per CU -- the kernel is bound by its stores; the transforms of one workgroup hide
// under the stores of the other), 64 rows for 19 and 21 frames (one workgroup per CU).
constexpr int kSRingPitch = 268;
constexpr int kStackMaxN = 21;                        // odd framenum values of the reference's scripts: 7 .. 21
static_assert(kStackMaxN - 1 == 20, "oring_floats() keeps kStackMaxN - 1 mirror rows");
constexpr int stack_ring_rows(int n) { return 2 * (n / 2) + kBatch <= 32 ? 32 : 64; }
constexpr int smem_floats_stacked(int n) { return kXbufFloats + kAbufFloats + stack_ring_rows(n) * kSRingPitch + kTwFloats; }
static_assert(smem_floats_stacked(17) * 4 <= 64 * 1280, "stacked kinds with windows up to 17 frames no longer fit two workgroups per CU");
static_assert(2 * (kStackMaxN / 2) + kBatch <= stack_ring_rows(kStackMaxN), "spectrum ring too short for the context window");
// gfx950 hands out LDS in units of 1280 bytes; three workgroups per CU fit while each takes at
// most 42 units of the CU's 128.  The 13-coefficient kernels meet that exactly.
static_assert(smem_floats(13) * 4 <= 42 * 1280 && smem_floats(kMaxCep) * 4 <= 42 * 1280, "LDS carve no longer allows three workgroups per CU");
static_assert((smem_floats(13) + kSumsLdsFloats) * 4 <= 42 * 1280 && (smem_floats(kMaxCep) + kSumsLdsFloats) * 4 <= 42 * 1280,
              "the column sums no longer fit beside the rest at three workgroups per CU");
static_assert((kBatch - 1) * 512 + row_skew(kBatch - 1) + kRowPad <= kXbufFloats, "spectrum rows fit in the exchange buffer");

constexpr float kEps64 = 2.220446049250313e-16f;     // np.finfo(float).eps, mfcc-self5.py:131

// Output frames per tile: 19 batches of 16.  A tile's cost outside its batches (set-up, the halo
// frames of a neighbouring tile) is spread over that many frames; more would only coarsen the
// load balance between workgroups.
inline int tile_frames_for(int /*numcep*/) { return 19 * kBatch; }

struct Tile {
  int32_t utt;  // utterance index
  int32_t t0;   // first output frame (utterance-relative)
  int32_t nt;   // number of output frames
  int32_t pad;
};

// How the [filters x 257 bins] x [257 x 16 frames] filterbank product is cut for
// v_mfma_f32_4x4x1_16b_f32 (16 independent 4x4 outer products per instruction).  The filters form
// groups of up to four consecutive ones (four each, unless the filters are so wide that a group has to
// be split to fit a wave's slots); a group's weights are non-zero on a short range of bins, which starts at a multiple of four and is walked in QUADS of four
// k-steps = 16 bins.  In a k-step, block 4 a + b of the instruction (lanes 16 a + 4 b .. + 3) multiplies
// the group's four weights of ONE bin with that bin's values of frames 4 a .. 4 a + 3: k-step st of
// quad i takes bin k0 + 16 i + 4 b + st for b = 0..3, so the four blocks of a 16-lane row split the
// range between them (their sums meet by two row rotations) and a lane's four k-steps of a quad
// are 16 contiguous bytes of its frame's spectrum row.  Every wave owns some groups; its quads,
// group after group, fill SLOTS 0 .. nslots-1, each described by one word:
//   bits 0-5   first bin of the quad / 4            bit 6   first quad of its group
//   bit 7      last quad of its group               bits 8-10  filters in the group (1..4)
//   bits 11-16 first filter of the group            bits 17-20 group number (its DCT weights)
struct MelMap {
  int32_t nslots[4];
  uint32_t slot[4][kMelQuads];
};
constexpr uint32_t kSlotFirst = 1u << 6, kSlotLast = 1u << 7;

// Per-plan constants, resident in device memory; read with scalar loads where needed so
// that they do not occupy scalar registers across the batch loop.
struct PlanConst {
  const float* tw512;     // [512][2]  exp(-2 pi i m / 512)
  const float* melW;      // [4 waves][kMelQuads][16][4] A operands of the filterbank (filter weights * spec_scale): lane
                          //   index 4 b + i (= lane & 15), element st = weight of filter (group's first + i) at the bin of
                          //   block b in k-step st of the slot's quad
  const float* dctW;      // [kMelGroups][16][4] DCT*lifter weights of the chained product, times ln 2 (the kernel feeds
                          //   it log2 of the mel sums): lane index 4 b + i, element r = weight of (coefficient 4 b + i,
                          //   filter r of group g)
  const float* window;    // [512] or nullptr
  MelMap map;
  int32_t nmel;
  int32_t numcep;
  float pre_hi, pre_lo;   // pre-emphasis coefficient split in two floats (hi + lo ~ double value); 0, 0 = off
  int32_t spectrum;       // 0 power, 1 magnitude
  int32_t floor_mel;
  float spec_scale;       // 1/(4*nfft) for power, 1/2 for magnitude (both exact powers of two)
  float log_floor;        // log-spectrum output: ln(max(v, log_floor)) when > 0 (anal_recog_nn_liste.m:529)
  int32_t decim;          // 1: 512-point plan; 2: 256-point plan = the even bins of the 512-point transform
};

// ---- streaming (B live streams in lock step, one workgroup per stream and push) ------------------
constexpr int kStreamHalo = 2 * kHalo;  // static rows kept from earlier pushes (delta-delta reaches 4 back)

struct StreamStep {
  // audio segments: [2][B][cap] samples, statics: [2][B][rows][kMaxCep] floats (ping-pong halves)
  int32_t n_streams, cap, rows;
  int32_t src_half;      // half holding the previous segment / statics
  int32_t src_skip;      // samples of the previous segment to drop from its front
  int32_t keep;          // samples of the previous segment carried over
  int32_t n_new;         // samples appended per stream
  int32_t prev_new_rows; // static rows the previous step computed (its rows kStreamHalo .. +prev_new_rows)
  int32_t t0, nt;        // first new frame (segment-relative) and number of new frames
  int32_t tile_max;      // frames per tile of the fused kernel
  int32_t tiles_per_stream;
};

// What the streaming kind of the fused kernel needs besides the step: the chunk to append and where
// the finished rows go.  Frame numbers e0, lo, hi are relative to the first new static row.
struct StreamArgs {
  StreamStep step = {};
  const void* chunk = nullptr;  // [B][chunk_ld] samples
  int64_t chunk_ld = 0;
  float* rows_out = nullptr;    // [B][out_stream_stride] floats, rows of out_ld
  int64_t out_stream_stride = 0, out_ld = 0;
  int32_t e0 = 0, k = 0;        // first row to emit, number of rows
  int32_t lo = 0, hi = 0;       // frame range of the utterance known so far (clamp of the regression taps)
};

// New segment of stream b = [carried tail | chunk]; the last kStreamHalo static rows move to the
// front of the other half.  Called by all 256 threads of one workgroup (or by several workgroups
// that split the samples: i0 = first sample of the caller, stride = samples per sweep).
template <typename T>
__device__ __forceinline__ void stream_carry(const StreamStep& s, int b, T* seg, float* stat, const T* chunk, int64_t chunk_ld,
                                             int i0, int stride, bool move_rows, int tid) {
  const int dst_half = s.src_half ^ 1;
  T* const dst = seg + ((int64_t)dst_half * s.n_streams + b) * s.cap;
  const T* const src = seg + ((int64_t)s.src_half * s.n_streams + b) * s.cap + s.src_skip;
  const int total = s.keep + s.n_new;
  for (int i = i0; i < total; i += stride) dst[i] = i < s.keep ? src[i] : chunk[(int64_t)b * chunk_ld + (i - s.keep)];
  if (move_rows) {
    float* const sd = stat + ((int64_t)dst_half * s.n_streams + b) * s.rows * kMaxCep;
    const float* const ss = stat + ((int64_t)s.src_half * s.n_streams + b) * s.rows * kMaxCep + s.prev_new_rows * kMaxCep;
    for (int i = tid; i < kStreamHalo * kMaxCep; i += 256) sd[i] = ss[i];
  }
}

// The regression of mfcc-self5.py:169-179 for pad_fac = 2: (-2a - b + 0m + d + 2e) / 10.  The zero-weight
// centre tap keeps NaN propagation identical to np.dot over the window.  Written with explicit
// roundings so that every caller (the fused kernel's scalar and packed forms, the streaming rows)
// gets the same bits: left to the compiler, the final product may or may not be fused into the
// caller's next subtraction.
__device__ __forceinline__ float regress5(float a, float b, float m, float d, float e) {
#pragma clang fp contract(off)  // the operations below carry no licence to be fused with a caller's, wherever this is inlined
  const float ea = e - a, db = d - b;
  const float t = fmaf(0.f, m, fmaf(2.f, ea, db));
  return t * 0.1f;
}

// One output row element: [c | delta | delta-delta] of frame e for coefficient c, the regression of
// mfcc-self5.py:169-179 applied twice.  S = the stream's static rows, row 0 = first new frame of
// this step, column c; taps are clamped to [lo, hi]: lo = the utterance's frame 0, hi = its last
// frame at flush (edge replication, mfcc-self5.py:174) or out of reach before.
__device__ __forceinline__ void stream_emit_one(const float* S, int e, int lo, int hi, int ncep, int c, float* o) {
  auto st = [&](int u) { return S[min(max(u, lo), hi) * kMaxCep]; };
  auto dl = [&](int v) {
    v = min(max(v, lo), hi);
    return regress5(st(v - 2), st(v - 1), st(v), st(v + 1), st(v + 2));
  };
  o[c] = st(e);
  o[ncep + c] = dl(e);
  o[2 * ncep + c] = regress5(dl(e - 2), dl(e - 1), dl(e), dl(e + 1), dl(e + 2));
}

struct FusedParams {
  const void* audio;
  int64_t total_elems;        // elements in `audio`, >= 1 (bounds the 16-byte staging loads)
  const int64_t* sample_off;  // [n_utts + 1]
  const int64_t* sample_len;  // [n_utts] or nullptr = sample_off[u+1] - sample_off[u]
  const int64_t* frame_off;   // [n_utts + 1]
  const Tile* tiles;
  float* out;
  int64_t out_ld;
  const PlanConst* pc;
  int32_t stack_n;            // stacked kinds: frames per context window N (odd, <= kStackMaxN); out_ld = N * bins
  float* tile_sums;           // kOutMfcc39Sums: [n_tiles][kTileSums] column sums of each tile's own rows
  const Tile* wg_tiles;       // split_kernel.h: the tiles of its persistent workgroups, workgroup after workgroup
  const int32_t* wg_off;      //   [workgroups + 1] first tile of every workgroup in wg_tiles
  StreamArgs stream;          // streaming kind only
#ifdef MFCC_DIAG
  unsigned long long* dbg;    // [n_tiles][4 waves][8] cycle sums
#endif
};

#ifdef MFCC_DIAG
// Diagnostic build only: per-wave cycle sums of the phases of the batch loop, written to a
// buffer of their own (never to an output element).
#define MFCC_STAMP(var) unsigned long long var; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var) :: "memory")
#else
#define MFCC_STAMP(var)
#endif

// Phase marks for tools/isa_phases.py (build with -DMFCC_PHASE_MARKS, assembly only): a comment line in the
// instruction stream wherever a phase of the batch loop begins.  They may shift the compiler's schedule a
// little, so the product is built without them.
#ifdef MFCC_PHASE_MARKS
#define MFCC_PHASE(name)                    \
  do {                                      \
    __builtin_amdgcn_sched_barrier(0);      \
    asm volatile("; MFCC_PHASE " name);     \
    __builtin_amdgcn_sched_barrier(0);      \
  } while (0)
#else
#define MFCC_PHASE(name)
#endif

enum : int { kOutMfcc39 = 0, kOutStatic = 1, kOutLogmel = 2, kOutSpec = 3, kOutLogspec = 4,
             kOutSpecStack = 5, kOutLogspecStack = 6,    // spectrum rows stacked over N frames (context windows)
             // One push of B live streams in one launch, workgroup b = stream b: append the chunk to the
             // stream's segment, static rows of the new complete frames (as kOutStatic, into the stream's
             // row store), then the output rows whose +-4-frame context is complete.
             kOutStream = 7,
             // MFCC-39 rows stacked over N frames, out[t, f * 3 ncep + j] = row[clamp(t + f - N / 2)][j] (padUtterance +
             // slidingWin with coeffs = 39, MLP_5hidlayer_40.py:70-105): the finished rows pass through a ring in
             // LDS, from which whole output rows are written
             kOutMfcc39Stack = 8,
             // kOutMfcc39 that also sums the rows it writes and their squares per column (the statistics of a corpus
             // normalisation, (x + mm) .* vv of anal_recog_nn_liste.m:180): every lane adds its three values of a row
             // to its wave's column sums in LDS; a tile leaves 6 x 16 float sums in FusedParams::tile_sums
             kOutMfcc39Sums = 9 };

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Pointers read out of PlanConst are generic to the compiler; the tables live in device
// memory, and saying so turns flat loads (which also tie up the LDS counter) into global ones.
typedef const float __attribute__((address_space(1))) * gfloat_ptr;
__device__ __forceinline__ gfloat_ptr as_global(const float* p) { return (gfloat_ptr)(uintptr_t)p; }

__device__ __forceinline__ void wave_sync() {
  // LDS operations of one wave execute in program order; this only stops the compiler
  // from moving LDS accesses across the exchange point.
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ln x from the hardware log2 (v_log_f32, 1 ulp): relative error ~2e-7 for the mel energies
// (x >> denormal range); ln 0 = -inf as numpy's.
__device__ __forceinline__ float fast_ln(float x) { return __builtin_amdgcn_logf(x) * 0.69314718055994531f; }

// 32-bit LDS address of a pointer into the workgroup's shared memory (for hand-written DS instructions)
__device__ __forceinline__ unsigned lds_addr(const void* p) {
  return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) void*)p;
}

// v of lane (16 - q) & 15 of the caller's row of 16 lanes, q = its own index in the row, by two data-parallel moves on the
// vector ALU: rotate the row left by one lane (row_ror:15), then mirror it
__device__ __forceinline__ float partner_dpp(float v) {
  const int r = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x120 + 15, 0xf, 0xf, true);
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, r, 0x140 /* row_mirror */, 0xf, 0xf, true));
}
__device__ __forceinline__ float lane_read(int byte_addr, float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(byte_addr, __builtin_bit_cast(int, v)));
}

// v[lane] of the lane N places further on in its row of 16 lanes (row_ror:N): adding the results for
// N = 8, 4 (and 2, 1) leaves every lane with the sum over the lanes of its row that differ in those bits.
template <int N>
__device__ __forceinline__ float row_ror(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x120 + N, 0xf, 0xf, true));
}

// Staging of one 16-frame batch into LDS as float, pre-emphasised, zero beyond the utterance
// end (mfcc-self5.py:68 and :89-90).  Global memory is read in aligned 16-byte chunks; the LDS
// image keeps the chunk alignment, i.e. sample i of the batch sits at abuf[delta + i] where
// delta = (address of the batch's first sample) mod 16 bytes, in elements.  The global loads
// (stage_load) are issued a phase before their results are converted and written
// (stage_store), so that the HBM latency is covered by the matrix-core work in between.
struct StageRegs {
  uint4 raw[3];   // chunks per thread: ceil((7 + 2800) / 8 / 256) = 2 for int16, 3 for float32
  float prev[3];  // sample in front of each chunk: float32 input as it is, int16 input as the integer's bits (converted
                  // by stage_store: a conversion here would make the loading code wait for its own loads)
  int delta;
};

// stage_load has no branch at all: addresses are clamped instead.  An aligned 16-byte chunk
// that holds at least one byte of the buffer cannot fault, so the only chunks to avoid are
// those entirely behind the buffer end (they are pulled back to the last valid chunk) and a
// thread without a chunk of its own repeats the batch's last chunk (same cache line, no extra
// HBM traffic).  Whatever such loads return lies outside the utterance and is overwritten with
// zeros by stage_store.  Being branch-free matters: the loads stay in one basic block with the
// matrix-core section that follows, which lets the compiler wait for the filterbank operands
// without also waiting for these (younger, HBM-latency) loads.
// Per-tile constants of the staging loads.  All batches of a tile share the alignment offset
// delta (16 frames advance by a multiple of 16 bytes), so the clamps are 32-bit element offsets
// from the tile's first aligned chunk, set up once; a batch only adds its own offset.
template <typename T>
struct StageCtx {
  const T* base;   // aligned chunk 0 of the tile's first batch
  int hi_rel;      // offset of the last aligned chunk that still holds a byte of the buffer (>= 0)
  int last_rel;    // offset of the last element of the buffer
  int front_rel;   // offset of the first element of the buffer (<= 0 except by misalignment)
  int delta;
  unsigned nchunk;
};

template <typename T, class G>
__device__ __forceinline__ StageCtx<T> stage_setup(const T* buf, int64_t total, int64_t s_off, int t_first) {
  // buf/total = the whole audio buffer (total >= 1), s_off = first sample of the utterance,
  // t_first = first frame the tile computes (its first sample lies inside the buffer)
  constexpr int CE = 16 / (int)sizeof(T);
  constexpr int64_t kBig = 1 << 28;
  static_assert((16 * G::kStep * sizeof(T)) % 16 == 0, "batches must keep the 16-byte phase");
  const T* const b0 = buf + s_off + (int64_t)t_first * G::kStep;
  StageCtx<T> c;
  c.delta = (int)(((uintptr_t)b0 & 15) / sizeof(T));
  c.base = b0 - c.delta;
  const T* const last = buf + (total - 1);
  const T* const hi_chunk = last - (((uintptr_t)last & 15) / sizeof(T));
  const int64_t hi64 = hi_chunk - c.base, last64 = last - c.base, front64 = buf - c.base;
  c.hi_rel = (int)(hi64 < kBig ? hi64 : kBig);
  c.last_rel = (int)(last64 < kBig ? last64 : kBig);
  c.front_rel = (int)(front64 > -kBig ? front64 : -kBig);
  c.nchunk = (c.delta + G::kStageLen + CE - 1) / CE;
  return c;
}

template <typename T, class G>
__device__ __forceinline__ void stage_load(const StageCtx<T>& c, StageRegs& r, int batch_off, int tid) {
  // batch_off = (first frame of the batch - first frame of the tile) * step, in elements
  constexpr int NR = sizeof(T) == 2 ? 2 : 3;
  r.delta = c.delta;
  // Byte offsets from a wave-uniform base, kept non-negative so that each load is one instruction
  // with a scalar base and a 32-bit vector offset (no 64-bit address arithmetic per lane).  The
  // predecessor sample may lie up to 14 bytes in front of the tile's first chunk: its offsets are
  // taken from 16 bytes below the base.
  constexpr int SZ = (int)sizeof(T);
  const char* const base_b = (const char*)c.base;
  const int hi_b = c.hi_rel * SZ, front_b = c.front_rel * SZ + 16, last_b = c.last_rel * SZ + 16;
#pragma unroll
  for (int k = 0; k < NR; ++k) {
    const int eb = batch_off * SZ + (int)min((unsigned)(tid + 256 * k), c.nchunk - 1) * 16;
    r.raw[k] = *(const uint4*)(base_b + (unsigned)min(eb, hi_b));
    const T pv = *(const T*)(base_b - 16 + (unsigned)min(max(eb + 16 - SZ, front_b), last_b));
    if (sizeof(T) == 2) r.prev[k] = __builtin_bit_cast(float, (int)pv);
    else r.prev[k] = (float)pv;
  }
}
// The same loads for the batches after a tile's first, through a buffer descriptor: its range check does
// the work of the clamps (a chunk that starts behind the last chunk holding a byte of the buffer reads as
// zeros instead of being pulled back; such samples lie outside the utterance and stage_store zeroes them
// anyway), and nothing of these batches lies in front of the buffer, so the address of a load is a
// per-lane offset that never changes plus wave-uniform scalars: no vector arithmetic per batch.
// The descriptor is rebuilt per batch from scalars (base moved on by the batch's offset, the record count
// shrunk by it), since the hardware leaves a scalar offset out of the range check.
template <typename T>
struct StageDesc {
  const char* base16;  // 16 bytes in front of the tile's first aligned chunk
  unsigned nrec;       // bytes from base16 to the end of the last chunk that holds a byte of the buffer
  unsigned voff[sizeof(T) == 2 ? 2 : 3];  // per lane: byte offset of its chunks inside a batch
};
template <typename T, class G>
__device__ __forceinline__ StageDesc<T> stage_desc(const StageCtx<T>& c, int tid) {
  constexpr int NR = sizeof(T) == 2 ? 2 : 3;
  StageDesc<T> d;
  d.base16 = (const char*)c.base - 16;
  d.nrec = (unsigned)c.hi_rel * (unsigned)sizeof(T) + 32u;
#pragma unroll
  for (int k = 0; k < NR; ++k) d.voff[k] = min((unsigned)(tid + 256 * k), c.nchunk - 1) * 16u;
  return d;
}
template <typename T, class G>
__device__ __forceinline__ void stage_load_desc(StageDesc<T>& d, const StageCtx<T>& c, StageRegs& r, int batch_off) {
  // batch_off = (first frame of the batch - first frame of the tile) * step, in elements, > 0
  constexpr int NR = sizeof(T) == 2 ? 2 : 3;
  constexpr unsigned SZ = (unsigned)sizeof(T);
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  r.delta = c.delta;
  const int boff = batch_off * (int)SZ;
  // (signed scalar arithmetic: an unsigned saturating difference would be formed on the vector unit and make
  // the descriptor a per-lane value; nrec < 2^31)
  constexpr int CE = 16 / (int)sizeof(T);
  constexpr int kMinChunks = (G::kStageLen + CE - 1) / CE;  // chunks of a batch at least
#pragma unroll
  for (int k = 0; k < NR; ++k) {
    // A round of chunks that every batch has in full (the second of three with float32 samples) needs no offset
    // register of its own: its chunks lie 4096 k bytes behind the first round's, and the descriptor moves instead.
    const bool derived = k > 0 && 256 * (k + 1) <= kMinChunks;
    const int shift = derived ? 4096 * k : 0;
    const int left = max((int)d.nrec - boff - shift, 0);
    const __amdgpu_buffer_rsrc_t rs =
        __builtin_amdgcn_make_buffer_rsrc((void*)const_cast<char*>(d.base16 + boff + shift), 0, left, 0x00020000);
    // (the offset passes through an empty asm statement, in its own register: the constants below then stay
    // immediate offsets of the loads instead of being added into registers of their own in front of the loop)
    unsigned& vreg = d.voff[derived ? 0 : k];
    asm volatile("" : "+v"(vreg));
    const unsigned vo = vreg;
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, vo + 16u, 0, 0);
    r.raw[k] = make_uint4(v.x, v.y, v.z, v.w);
    if (sizeof(T) == 2) r.prev[k] = __builtin_bit_cast(float, (int)(short)__builtin_amdgcn_raw_buffer_load_b16(rs, vo + (16u - SZ), 0, 0));
    else r.prev[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, vo + (16u - SZ), 0, 0));
  }
}

template <typename T, class G>
__device__ __forceinline__ void stage_store(const StageRegs& r, float* abuf, int64_t len, int tb, int tid,
                                            float pre_hi, float pre_lo) {
  constexpr int CE = 16 / (int)sizeof(T);
  constexpr int NR = sizeof(T) == 2 ? 2 : 3;
  constexpr int kBig = 1 << 20;
  const int delta = r.delta;
  const int64_t rem64 = len - (int64_t)tb * G::kStep;  // samples of this utterance from the batch start on
  const int i_hi = delta + (rem64 > kBig ? kBig : (int)rem64);
  const int nchunk = (delta + G::kStageLen + CE - 1) / CE;
  // interior batch: no utterance start, no utterance end inside the staged range
  const bool interior = tb > 0 && i_hi >= nchunk * CE;
#pragma unroll
  for (int k = 0; k < NR; ++k) {
    const int c = tid + 256 * k;
    if (c >= nchunk) break;
    const int e0 = c * CE;
    float x[CE + 1];
    x[0] = sizeof(T) == 2 ? (float)__builtin_bit_cast(int, r.prev[k]) : r.prev[k];
    const unsigned w[4] = {r.raw[k].x, r.raw[k].y, r.raw[k].z, r.raw[k].w};
    if (sizeof(T) == 2) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        x[1 + 2 * j] = (float)(int16_t)(w[j] & 0xffff);
        x[2 + 2 * j] = (float)((int)w[j] >> 16);
      }
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) x[1 + j] = __uint_as_float(w[j]);
    }
    float y[CE];
    // y[n] = x[n] - c x[n-1] (mfcc-self5.py:68).  int16 input: numpy works in float64, matched
    // by a two-term coefficient and fused multiply-adds, two samples per packed instruction;
    // float32 input: numpy works in float32 with two roundings.
    if (sizeof(T) == 2) {
#ifdef MFCC_STAGE_PACKED
      const cf nhi = mk(-pre_hi, -pre_hi), nlo = mk(-pre_lo, -pre_lo);
#pragma unroll
      for (int j = 0; j < CE; j += 2) {
        const cf x0 = mk(x[j], x[j + 1]), x1 = mk(x[j + 1], x[j + 2]);
        const cf r = __builtin_elementwise_fma(nlo, x0, __builtin_elementwise_fma(nhi, x0, x1));
        y[j] = r.x;
        y[j + 1] = r.y;
      }
#else
      // (plain fmas: the packed form needs the shifted pairs (x[j], x[j+1]) and (x[j+1], x[j+2]) put together with
      // a move per sample, and a packed fma costs more issue time than two plain ones save)
#pragma unroll
      for (int j = 0; j < CE; ++j) y[j] = __builtin_fmaf(-pre_lo, x[j], __builtin_fmaf(-pre_hi, x[j], x[j + 1]));
#endif
    } else {
#pragma unroll
      for (int j = 0; j < CE; ++j) y[j] = __fsub_rn(x[j + 1], __fmul_rn(pre_hi, x[j]));
    }
    if (!interior) {
#pragma unroll
      for (int j = 0; j < CE; ++j) {
        const int e = e0 + j;
        if (tb == 0 && e == delta) y[j] = x[j + 1];  // y[0] = x[0], mfcc-self5.py:68
        if (e >= i_hi) y[j] = 0.f;                   // zero tail, mfcc-self5.py:89-90
      }
    }
#pragma unroll
    for (int j = 0; j < CE; j += 4) *(float4*)(abuf + e0 + j) = make_float4(y[j], y[j + 1], y[j + 2], y[j + 3]);
  }
}

// NCEP > 0 fixes numcep at compile time (cheap index arithmetic in the epilogue); 0 = runtime.
// T is the sample type of the audio buffer (int16_t or float), G the frame geometry.
template <int OUT, int NCEP, typename T, class G>
__global__ void __launch_bounds__(256, MFCC_MIN_WAVES) fused_kernel(const FusedParams p) {
  extern __shared__ __attribute__((aligned(128))) float smem[];
  float* const xbuf = smem;
  float* const abuf = xbuf + kXbufFloats;
  // floats per row of the statics ring; only the MFCC-39 kind has one (the others would carry 2.5 KB
  // of LDS they never touch, one allocation unit too many for three workgroups per CU)
  constexpr bool kSums = OUT == kOutMfcc39Sums;
  constexpr bool kRows39 = (OUT == kOutMfcc39 || OUT == kOutMfcc39Stack || kSums);  // kinds that form delta and delta-delta rows
  constexpr int RP = kRows39 ? (NCEP > 0 ? NCEP : kMaxCep) : 0;
  constexpr bool kStack = (OUT == kOutSpecStack || OUT == kOutLogspecStack);
  float* const ring = abuf + kAbufFloats;
  float* const sring = abuf + kAbufFloats;  // stacked kinds: ring of spectrum rows in place of part / ring
  constexpr bool kMirror = ring_rows(RP) > 32;  // ring rows 0..7 repeated behind row 31
  const int srow_mask = kStack ? stack_ring_rows(p.stack_n) - 1 : 0;  // stacked kinds: rows of the spectrum ring - 1
  const int stack_sign = (p.stack_n & 2) ? -1 : 1;  // N % 4 == 3: an output row's N * 257 floats are congruent to -1 mod 4
  float* const twl = kStack ? sring + (srow_mask + 1) * kSRingPitch : ring + ring_rows(RP) * RP;
  float* const dctl = twl + kTwFloats;  // DCT * lifter weights (not in the stacked spectrum kinds' carve)
  float* const oring = dctl + kDctFloats;  // kOutMfcc39Stack: ring of finished rows
  float* const csum = dctl + kDctFloats;   // kOutMfcc39Sums: [4 waves][kTileSumCols] column sums, then K[16]
  float* const kpiv = csum + 4 * kTileSumCols;

  constexpr bool kSpecOut = (OUT == kOutSpec || OUT == kOutLogspec || kStack);
  constexpr bool kLogOut = (OUT == kOutLogspec || OUT == kOutLogspecStack);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = lane & 15;
  const int g4 = lane >> 4;
  const int fb = wave * 4 + g4;  // frame slot inside the batch

  const PlanConst* const pc = p.pc;
  Tile tile;
  int64_t s_off, len, f_off;
  int nframes;
  if (OUT == kOutStream) {
    // the step's layout follows from its scalars: segment and row store of stream b in the half
    // this push writes; row (f_off + t) of the row store holds segment frame t
    const StreamStep& ss = p.stream.step;
    const int b = blockIdx.x;
    stream_carry<T>(ss, b, (T*)const_cast<void*>(p.audio), p.out, (const T*)p.stream.chunk, p.stream.chunk_ld, tid, 256, true, tid);
    __syncthreads();  // the segment is read back below by other waves of this workgroup
    const int64_t slot = (int64_t)(ss.src_half ^ 1) * ss.n_streams + b;
    tile = Tile{b, ss.t0, ss.nt, 0};
    s_off = slot * ss.cap;
    len = ss.keep + ss.n_new;
    f_off = slot * ss.rows + kStreamHalo - ss.t0;
    nframes = ss.t0 + ss.nt;
  } else {
    tile = p.tiles[blockIdx.x];
    s_off = p.sample_off[tile.utt];
    len = p.sample_len ? p.sample_len[tile.utt] : p.sample_off[tile.utt + 1] - s_off;
    f_off = p.frame_off[tile.utt];
    nframes = (int)(p.frame_off[tile.utt + 1] - f_off);
  }
  const int t0 = tile.t0, nt = tile.nt;
  // frames this tile computes: its own, plus what the rows at its ends reach for where the utterance
  // continues (4 frames for delta-delta, half a context window for the stacked kinds)
  // (a stacked row also completes the 16-byte chunk it shares with the row before it: one frame back at least)
  const int reach = (OUT == kOutMfcc39 || kSums) ? kHalo : (OUT == kOutMfcc39Stack ? kHalo + (p.stack_n >> 1) : (kStack ? p.stack_n >> 1 : 0));
  const int s_lo = (reach || kStack) ? max(0, t0 - max(reach, kStack ? 1 : 0)) : t0;
  const int s_hi = (reach || kStack) ? min(nframes, t0 + nt + reach) : t0 + nt;
  const int nmel = pc->nmel;
  const int ncep = NCEP > 0 ? NCEP : pc->numcep;
  // kOutMfcc39Stack: rows o_lo .. o_hi - 1 of the utterance pass through the ring of finished rows
  const int o_lo = OUT == kOutMfcc39Stack ? max(0, t0 - (p.stack_n >> 1)) : 0;
  const int o_hi = OUT == kOutMfcc39Stack ? min(nframes, t0 + nt + (p.stack_n >> 1)) : 0;
  const int ld = (int)p.out_ld;
  float* const out_t0 = p.out + (f_off + t0) * p.out_ld;  // row t0 of this tile; 32-bit offsets from here
  const float pre_hi = pc->pre_hi, pre_lo = pc->pre_lo;
  const bool mag = pc->spectrum != 0;
  // plan constants of the spectrum kinds, read here, before any store of the kernel: inside the batch
  // loop they would be vector loads behind the staging loads on the in-order memory counter
  const float spec_scale = pc->spec_scale, spec_floor = kSpecOut ? pc->log_floor : 0.f;
  const bool half_band = pc->decim == 2;  // a 256-point plan: the even bins of the 512-point transform
  const gfloat_ptr window = as_global(pc->window);
  // this wave's share of the filterbank product: its slots of four k-steps (see MelMap), wave-uniform
  const int mel_nslots = pc->map.nslots[wave];
  unsigned slot_info[kMelQuads];
#pragma unroll
  for (int s = 0; s < kMelQuads; ++s) slot_info[s] = pc->map.slot[wave][s];
  typedef const f32x4 __attribute__((address_space(1))) * gfloat4_ptr;
  const gfloat4_ptr melW4 = (gfloat4_ptr)(uintptr_t)pc->melW + (wave * kMelQuads) * 16 + (lane & 15);

  // The first batch's samples are requested before the tables below are copied: their way from HBM is the longest
  // wait of a tile's set-up, and the table copies (L2 hits) run inside it.
  StageRegs sr;
  const StageCtx<T> sctx = stage_setup<T, G>((const T*)p.audio, p.total_elems, s_off, s_lo);
  stage_load<T, G>(sctx, sr, 0, tid);

  // Twiddles per lane index q: stage 1 uses W256^(q k1) = W512^(2 q k1), k1 = 1..15, the
  // untangle W512^(q + 16 s), s = 0..7, as (tan, cos) pairs.  They live in LDS (row q, 23 pairs) and are
  // read when needed: 46 registers per lane would cost a wave of occupancy.
  {
    const gfloat_ptr tw = as_global(pc->tw512);
    for (int i = tid; i < 16 * 23; i += 256) {
      const int qq = i / 23, k = i - qq * 23;
      const int m = k < 15 ? (2 * qq * (k + 1)) & 511 : 512 + qq + 16 * (k - 15);  // untangle entries: (tan, cos), behind the 512 powers
      const int slot = k < 15 ? kTwSt1 + k : kTwUnt + (k - 15);
      *(cf*)(twl + qq * kTwPitch + 2 * slot) = mk(tw[2 * m], tw[2 * m + 1]);
    }
  }
  if (!kSpecOut) {
    const gfloat_ptr dwg = as_global(pc->dctW);
    for (int i = tid; i < kDctFloats; i += 256) dctl[i] = dwg[i];
  }
  if (kSums)
    for (int i = tid; i < kSumsLdsFloats; i += 256) csum[i] = 0.f;
  const cf* const twq = (const cf*)(twl + q * kTwPitch);
#ifdef MFCC_PARTNER_BPERMUTE
  const int partner_addr = ((lane & 48) | partner_lane(q)) << 2;
#endif

#ifdef MFCC_STAGE_CLAMPED
  auto stage_ld_next = [&](int tb) { stage_load<T, G>(sctx, sr, (tb - s_lo) * G::kStep, tid); };
#else
  StageDesc<T> sdesc = stage_desc<T, G>(sctx, tid);
  auto stage_ld_next = [&](int tb) { stage_load_desc<T, G>(sdesc, sctx, sr, (tb - s_lo) * G::kStep); };  // tb > s_lo
#endif
  auto stage_st = [&](int tb) -> int {
    stage_store<T, G>(sr, abuf, len, tb, tid, pre_hi, pre_lo);
    return sr.delta;
  };
  // floor of the mel energies: 0 = none (ln 0 = -inf as the reference, mfcc-self5.py:125-127)
  const bool floor_mel = pc->floor_mel != 0;

  int delta = stage_st(s_lo);
  __syncthreads();

  // ---- delta, delta-delta and the output rows (mfcc-self5.py:169-186) ------------------------
  // (-2a - b + 0m + d + 2e) / 10; the zero-weight centre tap keeps NaN propagation identical to
  // np.dot over the window
  auto regress = [](float a, float b, float m, float d, float e) { return regress5(a, b, m, d, e); };
  // Rows r_first .. r_first + 15 (those inside the tile): wave w, lane group g4 take row
  // r_first + 4 w + g4, lane q its coefficient q, and write c | delta | delta-delta of that row, so
  // the three stores of a wave cover four whole consecutive output rows.  The statics of rows
  // r-4 .. r+4 come from the ring: consecutive ring rows away from the utterance's ends, row
  // indices clamped to the utterance near them (np.pad(mode='edge'), mfcc-self5.py:174).  The
  // reference pads twice -- delta-delta is the regression over the edge-padded DELTA sequence --
  // so a delta tap that falls outside the utterance takes the delta of the edge row, not the
  // regression over a shifted window.
  const unsigned row_lane_off = (unsigned)(g4 * ld + q);  // a lane's element inside its wave's four output rows
  // LDS address of the lane's column in ring row 0, opaque to the compiler: the ring's offset inside the carve
  // then stays part of this register instead of being added to every row address again
  unsigned ring_q_addr = lds_addr(ring + q);
  asm volatile("" : "+v"(ring_q_addr));
  auto emit_rows = [&](int r_first) {
    const int r = r_first + 4 * wave + g4;
    const int r_wave = r_first + 4 * wave;  // wave-uniform
    const bool interior = r_wave >= kHalo && r_wave + 3 + kHalo <= nframes - 1;
    float v[9], d[5];
    if (interior) {
      // two regressions per packed instruction: the rows are read once as the pairs (v0,v1) .. (v6,v7)
      // and once more as (v1,v2) .. (v7,v8); same operations in the same order as regress()
      const int b0 = r - kHalo - s_lo + 8;
      typedef const float __attribute__((address_space(3))) * lfloat_ptr;
      const lfloat_ptr w0 = (lfloat_ptr)(uintptr_t)(ring_q_addr + (unsigned)((b0 & 31) * (RP * 4)));
      auto at = [&](int m) { return kMirror ? w0[m * RP] : ring[((b0 + m) & 31) * RP + q]; };
      cf pe[4], po[4];
      if (kMirror) {
        // the odd pairs are read once more from LDS (through a pointer the compiler cannot tell from w0 + RP)
        // rather than put together from the even pairs' registers with six moves: the reads cost no vector issue
        int one_row = RP;
        asm volatile("" : "+s"(one_row));
        const lfloat_ptr w1 = w0 + one_row;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          pe[m] = mk(w0[2 * m * RP], w0[(2 * m + 1) * RP]);
          po[m] = mk(w1[2 * m * RP], w1[(2 * m + 1) * RP]);
        }
      } else {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          pe[m] = mk(at(2 * m), at(2 * m + 1));
          po[m] = mk(at(2 * m + 1), at(2 * m + 2));
        }
      }
      const cf two = mk(2.f, 2.f), zero = mk(0.f, 0.f), tenth = mk(0.1f, 0.1f);
      auto regress2 = [&](cf a, cf b, cf m, cf dd, cf e) {
        return __builtin_elementwise_fma(zero, m, __builtin_elementwise_fma(two, e - a, dd - b)) * tenth;
      };
      const cf d01 = regress2(pe[0], po[0], pe[1], po[1], pe[2]);  // rows k = 0, 1: taps k .. k + 4
      const cf d23 = regress2(pe[1], po[1], pe[2], po[2], pe[3]);
      v[4] = pe[2].x;
      d[0] = d01.x; d[1] = d01.y; d[2] = d23.x; d[3] = d23.y;
      d[4] = regress(pe[2].x, pe[2].y, pe[3].x, pe[3].y, po[3].y);
    } else {
      asm volatile("" ::: "memory");  // keeps this rare path a real branch: nothing of it is hoisted
      MFCC_PHASE("cold: rows near an utterance end");
#pragma unroll
      for (int m = 0; m < 9; ++m) {
        const int t = min(max(r - kHalo + m, 0), nframes - 1);
        v[m] = ring[((t - s_lo + 8) & 31) * RP + q];
      }
      float e[5];
#pragma unroll
      for (int k = 0; k < 5; ++k) e[k] = regress(v[k], v[k + 1], v[k + 2], v[k + 3], v[k + 4]);
      const int u = nframes - 1 - r;  // rows left above r; r itself = rows left below
      d[2] = e[2];
      d[1] = r == 0 ? e[2] : e[1];                          // row r-1 < 0: delta of row 0
      d[0] = r == 0 ? e[2] : (r == 1 ? e[1] : e[0]);        // row r-2 < 0
      d[3] = u == 0 ? e[2] : e[3];                          // row r+1 beyond the last row
      d[4] = u == 0 ? e[2] : (u == 1 ? e[3] : e[4]);
    }
    if (OUT == kOutMfcc39Stack) {
      if (q < ncep && r >= o_lo && r < o_hi) {
        const int slot = (r - o_lo) & (kORows - 1);
        float* const o = oring + slot * (3 * ncep) + q;
        const float dd = regress(d[0], d[1], d[2], d[3], d[4]);
        o[0] = v[4];
        o[ncep] = d[2];
        o[2 * ncep] = dd;
        if (slot < p.stack_n - 1) {  // mirror behind the ring's last row
          float* const om = o + kORows * (3 * ncep);
          om[0] = v[4];
          om[ncep] = d[2];
          om[2 * ncep] = dd;
        }
      }
    } else if (kSums) {
      // Column sums of the rows written.  Lane (g4, q) holds c, delta and delta-delta of row 4 wave + g4, column q:
      // the four rows of lanes are added first -- v_permlane32_swap / v_permlane16_swap exchange half-waves and
      // odd / even rows of two registers, so an exchange and an addition halve two values at once -- then the
      // wave adds the totals to its own sums in LDS (plain read, add, write: an LDS float atomic costs three
      // cycles per lane).  Columns >= numcep carry whatever the ring holds next to a row; nothing reads them.
      const float dd = regress(d[0], d[1], d[2], d[3], d[4]);
      const bool in = r >= t0 && r < t0 + nt;
      if (q < ncep && in) {
        float* const o = out_t0 + (int64_t)(r_wave - t0) * ld + row_lane_off;  // wave-uniform row + the lane's fixed offset
        o[0] = v[4];
        o[ncep] = d[2];
        o[2 * ncep] = dd;
      }
      if (r_wave + 3 >= t0 && r_wave < t0 + nt) {  // wave-uniform: some row of the wave belongs to the tile
        float c = v[4] - kpiv[q], dl = d[2], d2 = dd;
        if (r_wave < t0 || r_wave + 3 >= t0 + nt) {  // a tile's first or last rows: the others count as zero
          c = in ? c : 0.f;
          dl = in ? dl : 0.f;
          d2 = in ? d2 : 0.f;
        }
        // (written as instructions: the compiler folds the sum of the builtin's two results into twice the first;
        // a vector write of an operand has to be two wait states old when the exchange reads it)
        auto fold32 = [](float a, float b) {  // rows [a0 + a2, a1 + a3, b0 + b2, b1 + b3]
          asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
          return a + b;
        };
        auto fold16 = [](float a, float b) {  // rows [a0 + a1, b0 + b1, a2 + a3, b2 + b3]
          asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
          return a + b;
        };
        const float x = fold32(c, dl), y = fold32(d2, c * c), z = fold32(dl * dl, d2 * d2);
        const float r1 = fold16(x, y);  // rows: c, delta-delta, delta, c^2
        const float r2 = fold16(z, z);  // rows: delta^2 (twice), delta-delta^2 (twice)
        float* const cs = csum + wave * kTileSumCols;
        cs[lane] += r1;
        if (!(lane & 16)) cs[64 + 16 * (lane >> 5) + q] += r2;
      }
    } else if (q < ncep && r >= t0 && r < t0 + nt) {
      float* const o = out_t0 + (int64_t)(r_wave - t0) * ld + row_lane_off;  // wave-uniform row + the lane's fixed offset
      o[0] = v[4];
      o[ncep] = d[2];
      o[2 * ncep] = regress(d[0], d[1], d[2], d[3], d[4]);
    }
  };

  // ---- kOutMfcc39Stack: whole output rows from the ring of finished rows ---------------------------------
  // out[t, f OP + j] = row[clamp(t + f - h)][j], OP = 3 numcep, W = N OP floats per output row.  Rows
  // below rows_done are in the ring.  One wave writes a whole output row front to back as aligned 16-byte
  // chunks (chunk ch = the row's floats 4 ch - a .. 4 ch - a + 3, a = the row's offset from a 16-byte
  // boundary; a chunk belongs to the row that holds its last float, so the first chunk of a row also
  // carries the last a floats of the row before it).  Away from the utterance's ends the window's N rows
  // are one contiguous run in the ring; near them every float is looked up on its own.
  int stack_next = t0;  // stacked kinds: first output row not yet written
  auto write_stack_rows = [&](int rows_done) {
    const int N = p.stack_n, hN = N >> 1, OP = 3 * ncep, W = N * OP;
    const int t_end = min(rows_done >= nframes ? nframes : rows_done - hN, t0 + nt);
    auto row_at = [&](int sfrm) { return oring + ((min(max(sfrm, 0), nframes - 1) - o_lo) & (kORows - 1)) * OP; };
    for (int t = stack_next + wave; t < t_end; t += 4) {
      const int a = (int)(((f_off + t) * (int64_t)W) & 3);
      float* const orow = p.out + (f_off + t) * (int64_t)W;
      float* const obase = orow - a;  // 16-byte aligned
      const bool interior = t >= hN && t + hN <= nframes - 1;
      const int ch_hi = (W + a - 4) >> 2;
      const float* const win = oring + ((t - hN - o_lo) & (kORows - 1)) * OP - a;  // interior rows: float g of the row at win[a + g]
      for (int ch = lane; ch <= ch_hi; ch += 64) {
        f32x4 v;
        if (interior && ch > 0) {
          const float* const src = win + 4 * ch;
          v = f32x4{src[0], src[1], src[2], src[3]};
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            int g = 4 * ch - a + e, tt = t;
            if (g < 0) { g += W; tt = t - 1; }  // tail of the row before
            const int f = g / OP, j = g - f * OP;
            v[e] = row_at(tt + f - hN)[j];
          }
        }
        if (t == 0 && ch == 0 && a > 0) {  // first row of the utterance: the floats in front belong to another utterance
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (e >= a) obase[e] = v[e];
        } else {
          *(f32x4*)__builtin_assume_aligned(obase + 4 * ch, 16) = v;
        }
      }
      if (t == nframes - 1) {  // last row of the utterance: nobody follows to complete its last chunk
        const int ntail = (a + W) & 3;
        if (lane < ntail) orow[W - ntail + lane] = row_at(nframes - 1)[OP - ntail + lane];
      }
    }
    stack_next = max(stack_next, t_end);
  };

#ifdef MFCC_DIAG
  unsigned long long dsum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  MFCC_STAMP(tk0);
#endif
  for (int tb = s_lo; tb < s_hi; tb += kBatch) {
    MFCC_PHASE("1 frame load (staged samples -> registers)");
    MFCC_STAMP(ta);
    __builtin_amdgcn_s_setprio(MFCC_PRIO_FFT);
    // spectrum kinds have no filterbank phase to cover the staging loads of the next batch: they fly
    // during this batch's transforms instead
    if (kSpecOut && !kStack) stage_ld_next(tb + kBatch);
    f32x4 bv[kMelQuads];  // filter weights of this wave's slots (A operands), fetched at the end of the FFT section
    // ---- FFT-512 of 4 frames per wave, 16 lanes per frame -----------------------------
    {
      const float* fa = abuf + delta + fb * G::kStep;
      cf x[16];
      if (delta & 1) {  // odd sample offset: the (x[2n], x[2n+1]) pairs are not 8-byte aligned in LDS
        MFCC_PHASE("cold: frame load at an odd sample offset");
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
        // Registers that lie wholly inside the frame are read with one ds_read_b64 each, written out
        // by hand: the compiler pairs neighbouring 8-byte reads into ds_read2_b64, which occupies the
        // LDS twice as long for the same bytes.  Issued in the order the first pass consumes them
        // (x[b], x[4+b], x[8+b] for b = 0..3); LDS reads return in order, so a counted wait releases
        // one butterfly's operands at a time.  The waits carry the registers as operands so that no
        // use can be scheduled in front of them.
        constexpr int kFull = (G::kSize / 2) / 16;  // registers n1 < kFull need no mask
        const unsigned xa = lds_addr(fa2 + q);
        int issued = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
          for (int a = 0; a < 4; ++a) {
            const int n1 = 4 * a + b;
            if (n1 < kFull) {
              asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(x[n1]) : "v"(xa), "i"(n1 * 128));
              ++issued;
            }
          }
        (void)issued;
#pragma unroll
        for (int n1 = kFull; n1 < 16; ++n1) {
          const int n = 16 * n1 + q;
          if (2 * (16 * n1) < G::kSize) {
            const float ok = 2 * n + 1 < G::kSize ? 1.f : 0.f;  // staged samples are finite: 0 * v = 0
            x[n1] = fa2[min(n, G::kSize / 2 - 1)] * mk(ok, ok);
          } else {
            x[n1] = mk(0.f, 0.f);
          }
        }
        static_assert(kFull == 12 || kFull == 10 || kFull == 6, "wait counts below are written for 400-, 320- and 200-sample frames");
        if (kFull == 12) {
          asm volatile("s_waitcnt lgkmcnt(9)" : "+v"(x[0]), "+v"(x[4]), "+v"(x[8]));
          asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(x[1]), "+v"(x[5]), "+v"(x[9]));
          asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(x[2]), "+v"(x[6]), "+v"(x[10]));
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(x[3]), "+v"(x[7]), "+v"(x[11]));
        } else if (kFull == 10) {  // issued: 0 4 8 | 1 5 9 | 2 6 | 3 7
          asm volatile("s_waitcnt lgkmcnt(7)" : "+v"(x[0]), "+v"(x[4]), "+v"(x[8]));
          asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(x[1]), "+v"(x[5]), "+v"(x[9]));
          asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(x[2]), "+v"(x[6]));
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(x[3]), "+v"(x[7]));
        } else {
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]));
        }
      }
      if (window) {
        MFCC_PHASE("cold: window");
        // all window loads first (they hit L1/L2), then the products: one wait instead of 13
        cf wv[(G::kSize + 31) / 32];
#pragma unroll
        for (int n1 = 0; n1 < (G::kSize + 31) / 32; ++n1) wv[n1] = mk(window[2 * (16 * n1 + q)], window[2 * (16 * n1 + q) + 1]);
#pragma unroll
        for (int n1 = 0; n1 < (G::kSize + 31) / 32; ++n1) x[n1] = x[n1] * wv[n1];
      }
      // stage-1 twiddles of this lane: read now (their latency hides under the first 16-point
      // transform); reading them between the exchange writes would serialise on LDS round trips
      MFCC_PHASE("2 transform pass 1 (pruned 16-point, stage twiddles, exchange write)");
      cf tw1[16];
      {
        const float4* const tv = (const float4*)__builtin_assume_aligned(twq + kTwSt1, 16);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float4 v = tv[j];  // the last half of the last read is padding
          tw1[2 * j] = mk(v.x, v.y);
          tw1[2 * j + 1] = mk(v.z, v.w);
        }
      }
#ifndef MFCC_ABL_FFT
      fft16_pruned<(G::kSize + 31) / 32>(x);  // registers from ceil(size / 32) on hold zero padding
#endif
      // Exchange buffer of this frame: 2 KiB at byte offset fb*2048 (the LDS carve starts at a
      // 128-byte boundary), element Y[k1][n2] at xpose_index(k1, n2)*8.  Written out in bytes so
      // that each swizzled address is one XOR away from a per-lane base:
      //   write  k1*128 + ((q ^ (k1&14)) << 3)  =  (wbase ^ ((k1&14) << 3)) + k1*128
      //   read   q*128  + (((2j) ^ (q&14)) << 3) =  rbase ^ (j << 4)
      char* const lds0 = (char*)smem;
      const int wbase = fb * 2048 + (q << 3);
      const int rbase = fb * 2048 + (q << 7) + ((q & 14) << 3);
#pragma unroll
      for (int k1 = 0; k1 < 16; ++k1) {
#ifndef MFCC_ABL_FFT
        if (k1) x[k1] = cmul(x[k1], tw1[k1 - 1]);
#endif
        *(cf*)(lds0 + (wbase ^ ((k1 & 14) << 3)) + k1 * 128) = x[k1];
      }
      MFCC_PHASE("3 transform pass 2 (exchange read, 16-point)");
      wave_sync();
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float4 v = *(const float4*)(lds0 + (rbase ^ (j << 4)));
        x[2 * j] = mk(v.x, v.y);
        x[2 * j + 1] = mk(v.z, v.w);
      }
#ifndef MFCC_ABL_FFT
      fft16(x);
#endif
      wave_sync();  // every lane of this wave has read its row before spectrum rows overwrite it

      MFCC_PHASE("4 untangle, spectrum rows to LDS, energy sum");
      if (!kSpecOut) {
        // fetch the filterbank operands of this wave now (L2 latency ~ the untangle below)
#pragma unroll
        for (int s = 0; s < kMelQuads; ++s) bv[s] = melW4[s * 16];
      }
      // Stacked kinds keep the finished spectrum rows in a ring in LDS, each shifted by a phase of 0..3
      // floats so that the row writer below moves aligned chunks: element c of frame s lands at output float
      // (f_off + t) N B + f B + c with s = t + f - h, which for B = 257 is congruent mod 4 to
      // f_off + s + h + c when N % 4 == 1 and to s + h - f_off + c - 2 t when N % 4 == 3: with the phase
      // (+-f_off + s + h) mod 4 every chunk of the first case starts on a 16-byte boundary of the ring, and
      // those of the second on an 8-byte boundary (rows with odd t sit two floats off).
      float* const P = kStack ? sring + ((tb + fb - s_lo) & srow_mask) * kSRingPitch + (int)((stack_sign * f_off + tb + fb + (p.stack_n >> 1)) & 3)
                              : xbuf + fb * 512 + row_skew(fb);
      float* const Phi = P + (256 - 16 * 7) - q;  // the lane's upper bins 256 - q - 16 s, lowest address first
      // spectrum kinds: output row of this lane group's frame (nullptr past the tile's last frame)
      float* const spec_row = (kSpecOut && !kStack && tb + fb < s_hi) ? out_t0 + (tb + fb - t0) * ld : nullptr;
      auto spec_value = [&](float p4) {
        // LDS rows would hold 4|X|^2 (power) or, after the square root, 2|X| (magnitude)
        float v = (mag ? __builtin_amdgcn_sqrtf(p4) : p4) * spec_scale;
        if (kLogOut) v = fast_ln(fmaxf(v, spec_floor));  // floor 0 = none: ln 0 = -inf as np.log
        return v;
      };
      // the frame's energy sum (mfcc-self5.py:130) is collected on the way: every bin passes through
      // exactly one lane of the frame here
      cf esum = mk(0.f, 0.f);
      // Z[256-k] lives in the partner lane (16 - q) & 15 of the frame's row of lanes: four partners and four untangle
      // twiddles at a time
#pragma unroll
      for (int h = 0; h < 8; h += 4) {
        cf pb[4], pw[4];
#pragma unroll
#ifndef MFCC_PARTNER_BPERMUTE
        // (two vector moves per value instead of a ds_bpermute: the LDS path is the scarcer resource, 0.970 -> 0.952 ms)
        for (int s = 0; s < 4; ++s) pb[s] = mk(partner_dpp(x[15 - h - s].x), partner_dpp(x[15 - h - s].y));
#else
        for (int s = 0; s < 4; ++s) pb[s] = mk(lane_read(partner_addr, x[15 - h - s].x), lane_read(partner_addr, x[15 - h - s].y));
#endif
        {
          const float4* const tv = (const float4*)__builtin_assume_aligned(twq + kTwUnt + h, 16);
          const float4 v0 = tv[0], v1 = tv[1];
          pw[0] = mk(v0.x, v0.y); pw[1] = mk(v0.z, v0.w); pw[2] = mk(v1.x, v1.y); pw[3] = mk(v1.z, v1.w);
        }
        float lo4[4], hi4[4];  // MFCC kinds: the group's eight values, stored together below
#ifndef MFCC_Q0_SELECT
        // lane 0 pairs inside itself: Z[16 s] with Z[16 (16 - s)].  A real branch (the empty asm statement keeps it
        // one): four 64-bit moves under the lane mask instead of eight selects in the three-operand encoding
        if (q == 0) {
          asm volatile("");
#pragma unroll
          for (int s = 0; s < 4; ++s) pb[s] = x[(16 - h - s) & 15];
        }
#endif
#pragma unroll
        for (int s = 0; s < 4; ++s) {
#ifdef MFCC_Q0_SELECT
          if (q == 0) pb[s] = x[(16 - h - s) & 15];
#endif
          float lo, hi;
#ifndef MFCC_ABL_UNT
          untangle_power_tc(x[h + s], pb[s], pw[s], lo, hi);
#else
          lo = x[h + s].x + pb[s].x + pw[s].x; hi = x[h + s].y + pb[s].y + pw[s].y;
#endif
          if (kStack) {
            P[q + 16 * (h + s)] = spec_value(lo);
            P[256 - q - 16 * (h + s)] = spec_value(hi);
          } else if (kSpecOut) {
            // spectrum kinds: scale (sqrt, ln) and write the two bins of this lane straight from
            // registers to the output row of the frame; no trip through LDS, no barrier
            if (spec_row) {
              // a 256-point plan keeps the even bins (both bins of a lane have the parity of q)
              if (!half_band) {
                spec_row[q + 16 * (h + s)] = spec_value(lo);
                spec_row[256 - q - 16 * (h + s)] = spec_value(hi);
              } else if (!(q & 1)) {
                spec_row[(q >> 1) + 8 * (h + s)] = spec_value(lo);
                spec_row[128 - (q >> 1) - 8 * (h + s)] = spec_value(hi);
              }
            }
          } else {
            lo4[s] = lo;
            hi4[s] = hi;
          }
        }
        if (!kSpecOut) {
          if (mag) {
            // magnitude spectrum (mfcc-self15.py:234): one branch per group, so that the power path
            // does not pay for the square roots
            MFCC_PHASE("cold: magnitude square roots");
            asm volatile("" ::: "memory");
#pragma unroll
            for (int s = 0; s < 4; ++s) {
              lo4[s] = __builtin_amdgcn_sqrtf(lo4[s]);
              hi4[s] = __builtin_amdgcn_sqrtf(hi4[s]);
            }
          }
#pragma unroll
          for (int s = 0; s < 4; ++s) P[q + 16 * (h + s)] = lo4[s];
#pragma unroll
          for (int s = 0; s < 4; ++s) Phi[16 * (7 - (h + s))] = hi4[s];  // = P[256 - q - 16 (h + s)]: one base register and offsets
#pragma unroll
          for (int s = 0; s < 4; ++s) esum = esum + mk(lo4[s], hi4[s]);
        }
      }
      if (q == 0) {  // bin 128 pairs with itself: X[128] = conj Z[128]
        if (kStack) {
          P[128] = spec_value(4.f * norm2(x[8]));
        } else if (kSpecOut) {
          if (spec_row) spec_row[half_band ? 64 : 128] = spec_value(4.f * norm2(x[8]));
        } else {
          const float p128 = 4.f * norm2(x[8]);
          float v128 = p128;
          if (mag) {  // a branch on the wave-uniform flag: the power path pays for neither the root nor a select
            asm volatile("");
            v128 = __builtin_amdgcn_sqrtf(p128);
          }
          P[128] = v128;
          // (P[257..259], read with zero weights by the last k-steps, keep what this frame's transform left in its
          // exchange slot: finite whenever the frame is, and a frame that is not has no finite filter sum anyway)
          esum.x += v128;
        }
      }
      if (!kSpecOut) {
        // sum over the frame's 16 lanes; a 256-point plan counts the even bins only (a lane's bins have the parity of q)
        float e = esum.x + esum.y;
        if (half_band) {  // wave-uniform: a branch, not two selects in every plan
          asm volatile("");
          if (q & 1) e = 0.f;
        }
        e += row_ror<8>(e);
        e += row_ror<4>(e);
        e += row_ror<2>(e);
        e += row_ror<1>(e);
        if (q == 0) xbuf[fb * kSlot + kPartEnergy] = e;
      }
    }
#if !defined(MFCC_ABL_STAGE) && defined(MFCC_EARLY_STAGE)
    stage_ld_next(tb + kBatch);
    __builtin_amdgcn_sched_barrier(0);
#endif
    MFCC_PHASE("5 barrier 1");
    MFCC_STAMP(tb1);
    __syncthreads();  // spectrum rows complete; staged audio no longer needed
    MFCC_STAMP(tb2);
    __builtin_amdgcn_s_setprio(MFCC_PRIO_MEL);
    const bool more = tb + kBatch < s_hi;

    if (kStack) {
      // ---- context stacking (padUtterance + slidingWin, MLP_5hidlayer_40.py:70-105):
      // out[t, f B + c] = spec[clamp(t + f - h), c].  Row-driven: once the frames t - h .. t + h of an
      // output row exist in the ring, one wave writes the whole row (N B contiguous floats) front to
      // back with aligned 16-byte stores -- the order HBM takes best (a frame-by-frame scatter of
      // 1028-byte pieces with single floats at their ends measured 25 % slower).  Per block f: the
      // whole 16-byte chunks inside it come straight from the source row (aligned in LDS by its
      // phase); the chunk that straddles two blocks (or two rows) is put together from single floats
      // by one lane.  Only the first and last chunk of an utterance are partial.
      stage_ld_next(tb + kBatch);
      constexpr int B = kNbin;
      const int N = p.stack_n, hN = N >> 1;
      const int64_t W = (int64_t)N * B;
      float* const out_u = p.out + f_off * W;  // output row 0 of this utterance
      const int done = min(tb + kBatch, s_hi);  // frames 0 .. done-1 of the utterance are (or were) in the ring
      const int t_end = min(done == nframes ? nframes : done - hN, t0 + nt);
      typedef float f32x4_t __attribute__((ext_vector_type(4)));
      auto ring_row = [&](int sfrm) { return sring + ((sfrm - s_lo) & srow_mask) * kSRingPitch + (int)((stack_sign * f_off + sfrm + hN) & 3); };
      const bool half_aligned = stack_sign < 0;  // N % 4 == 3: chunks lie on 8-byte boundaries of the ring rows
      for (int t = stack_next + wave; t < t_end; t += 4) {
        const int a = (int)((stack_sign * (f_off + t)) & 3);  // floats of the row's start past a 16-byte boundary ((f_off + t) W mod 4)
        float* const orow = out_u + (int64_t)t * W;
        const bool interior = t >= hN && t + hN <= nframes - 1;  // no window index of this row is clamped
        // The chunks across block boundaries first: lane j puts together the one that ends with the
        // first c0 floats of block j (for j = 0 it begins in the last block of row t - 1).
        float jv[4] = {0.f, 0.f, 0.f, 0.f};
        if (lane < N) {
          const int j = lane, c0 = (4 - ((a + j) & 3)) & 3;
          if (c0 > 0) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int g = c0 - 4 + e;  // float index relative to the start of block j
              const int tt = (g < 0 && j == 0) ? t - 1 : t;
              const int blk = g < 0 ? (j == 0 ? N - 1 : j - 1) : j;
              const int c = g < 0 ? g + B : g;
              const int sc = min(max(tt + blk - hN, 0), nframes - 1);
              jv[e] = ring_row(sc)[c];
            }
          }
        }
        // The row as a run of 16-byte chunks, 64 per instruction: chunk ch covers the row's floats
        // 4 ch - a .. 4 ch - a + 3 and belongs to the row that holds its last float, so every store
        // instruction writes 1 KiB without a gap.  A chunk inside one block is one aligned 16-byte LDS
        // read of the source row; a chunk across two blocks comes from the lane that built it above.
        const int ch_hi = ((int)W + a - 4) >> 2;
        float* const obase = orow - a;  // 16-byte aligned
        // Four store instructions' worth of chunks at a time: their LDS reads are issued together, so
        // a wave keeps several stores in flight.  Two copies of the loop -- rows whose window is not
        // clamped (aligned 16-byte reads, no index clamps) and rows near the utterance's ends -- so
        // that no branch sits between the reads (a branch makes the compiler wait for each one).
        auto write_chunks = [&](auto interior_tag, auto half_tag) {
          constexpr bool kInterior = decltype(interior_tag)::value;
          constexpr bool kHalf = decltype(half_tag)::value;
          constexpr int U = 4;  // (2, 6 and 8 measure the same: 9.87 - 9.94 ms for the 52 GB of config C)
          for (int ch0 = 0; ch0 <= ch_hi; ch0 += 64 * U) {
            f32x4_t v[U];
            bool across[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
              const int ch = ch0 + 64 * u + lane;
              const int g0 = 4 * min(ch, ch_hi) - a;  // lanes past the row's end repeat its last chunk (and do not store)
              const int f = (int)(__umul24((unsigned)max(g0, 0), 16321u) >> 22);  // g0 / 257 for 0 <= g0 < 2^15
              const int c = g0 - f * B;
              across[u] = g0 < 0 || c > B - 4;
              // phase of the source row + c is a multiple of four floats for every chunk that starts inside
              // block f (a chunk across the boundary reads past bin 256 into the row's padding and is replaced
              // below); the one chunk that starts in front of the row reads an aligned dummy
              const int cr = g0 < 0 ? 4 - a : c;
#ifdef MFCC_ABL_ROWSRC
              v[u] = f32x4_t{(float)f, jv[0], (float)c, 0.f};
#else
              if (kInterior) {  // unclamped: frame t + f - h; c + its phase is congruent to c + a + f (+ 2 t when N % 4 == 3) mod 4
                const float* const src = ring_row(t + f - hN) + cr;
                if (kHalf) {
                  const float2 lo2 = *(const float2*)__builtin_assume_aligned(src, 8), hi2 = *(const float2*)__builtin_assume_aligned(src + 2, 8);
                  v[u] = f32x4_t{lo2.x, lo2.y, hi2.x, hi2.y};
                } else {
                  v[u] = *(const f32x4_t*)__builtin_assume_aligned(src, 16);
                }
              } else {
                const float* const src = ring_row(min(max(t + f - hN, 0), nframes - 1)) + cr;
                v[u] = f32x4_t{src[0], src[1], src[2], src[3]};
              }
#endif
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
              const int chu = ch0 + 64 * u;  // wave-uniform: first chunk of this instruction
              if (chu > ch_hi) break;
              // 64 chunks span 256 floats, fewer than a block: at most one of them lies across a block
              // boundary, the one in front of block jl (lane jl built it)
              const int g_first = 4 * chu - a;
              const int jl = g_first < 0 ? 0 : g_first / B + 1;
#if !defined(MFCC_ABL_ROWSRC) && !defined(MFCC_ABL_ROWJ)
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const float w = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, jv[e]), jl));
                v[u][e] = across[u] ? w : v[u][e];
              }
#endif
              const int ch = chu + lane;
              if (ch <= ch_hi) {
                if (!kInterior && t == 0 && ch == 0 && a > 0) {  // first row of the utterance: the floats in front belong to another utterance
#pragma unroll
                  for (int e = 0; e < 4; ++e)
                    if (e >= a) obase[e] = v[u][e];
                } else {
                  *(f32x4_t*)__builtin_assume_aligned(obase + 4 * ch, 16) = v[u];
                }
              }
            }
          }
        };
        // a first row (t == 0) is never interior except for a window of one frame, which the clamped copy handles too
        if (interior && t > 0) {
          if (half_aligned) write_chunks(std::true_type{}, std::true_type{});
          else write_chunks(std::true_type{}, std::false_type{});
        } else {
          write_chunks(std::false_type{}, std::false_type{});
        }
        if (t == nframes - 1) {  // last row of the utterance: nobody follows to complete its last chunk
          const int ntail = (a + (int)W) & 3;
          if (lane < ntail) orow[W - ntail + lane] = ring_row(nframes - 1)[B - ntail + lane];
        }
      }
      stack_next = max(stack_next, t_end);
      if (more) delta = stage_st(tb + kBatch);
      __syncthreads();
      continue;
    }
    if (kSpecOut) {
      // rows are already written (from registers, in the untangle step); the barrier above only released
      // the staged audio.  The loads of the next batch's samples were issued before this batch's
      // transforms: they are older than the row stores, so consuming them does not wait for stores (the
      // memory counter is in order), and the stores stay in flight across the barrier.
      if (more) delta = stage_st(tb + kBatch);
      __syncthreads();
      continue;
    }

    // ---- mel filterbank on the matrix cores: D[frame][col] = sum_bin P[frame][bin] W[bin][col]
#if !defined(MFCC_ABL_STAGE) && !defined(MFCC_EARLY_STAGE)
    MFCC_PHASE("6 next batch's global loads, filterbank operand reads");
    stage_ld_next(tb + kBatch);  // unconditional and branch-free; HBM latency runs under the MFMAs below
    __builtin_amdgcn_sched_barrier(0);  // keep the loads in front of the matrix-core chain
#endif
    {
      // Lane 16 a + 4 b + c: block 4 a + b of the matrix instruction, column c.  As B operand it brings the
      // spectrum of frame slot 4 a + c at the bin of block b; as A operand the weight of the group's filter c
      // at that bin; its four result registers are filters 0..3 of the group for frame slot 4 a + c, summed
      // over the bins block b walked.
      const int frm = (lane >> 4) * 4 + (lane & 3);
      const int blk = (lane >> 2) & 3;
      const float* Prow = xbuf + frm * 512 + row_skew(frm) + 4 * blk;
      const float* const dct_lane = dctl + 4 * (lane & 15);
      f32x4 pd = {0.f, 0.f, 0.f, 0.f};  // this wave's share of the frame's cepstra: coefficients 4 blk .. 4 blk + 3
#ifndef MFCC_ABL_MEL
      // All spectrum reads first (no branch between them, so they are issued ahead of the chain); a slot
      // the wave does not use reads bin 0.. with zero weights and is skipped below.
      f32x4 pa[kMelQuads];
#pragma unroll
      for (int s = 0; s < kMelQuads; ++s) {
        const float4 v = *(const float4*)__builtin_assume_aligned(Prow + 4 * (slot_info[s] & 63), 16);
        pa[s] = f32x4{v.x, v.y, v.z, v.w};
      }
      f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f}, dwc = {0.f, 0.f, 0.f, 0.f};
      // (the slot words pass through an empty asm statement so that the scalar tests on them are made here:
      // hoisted out of the batch loop they would occupy some fifty scalar registers, more than there are)
      int nslots_now = mel_nslots;
      asm volatile("" : "+s"(nslots_now));
#pragma unroll
      for (int s = 0; s < kMelQuads; ++s) {
        if (s >= nslots_now) break;  // wave-uniform
        unsigned info = slot_info[s];
        asm volatile("" : "+s"(info));
        MFCC_PHASE("7 filterbank slot (x23 per workgroup and batch, 5-6 per wave)");
        if (info & kSlotFirst) {
          // (starting the two chains from the instruction's constant zero operand in a branch of their own saves
          // these eight moves and measures 1.5 % slower: 1.025 against 1.009 ms)
          acc = f32x4{0.f, 0.f, 0.f, 0.f};
          acc2 = f32x4{0.f, 0.f, 0.f, 0.f};
          dwc = *(const f32x4*)__builtin_assume_aligned(dct_lane + 64 * ((info >> 17) & 15), 16);
        }
        // two accumulators: the dependency chain of the matrix instructions is half as long
        acc = __builtin_amdgcn_mfma_f32_4x4x1f32(bv[s][0], pa[s][0], acc, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_4x4x1f32(bv[s][1], pa[s][1], acc2, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_4x4x1f32(bv[s][2], pa[s][2], acc, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_4x4x1f32(bv[s][3], pa[s][3], acc2, 0, 0, 0);
        if (info & kSlotLast) {
          MFCC_PHASE("8 group end: reduce, log2, DCT k-steps (x7 per workgroup and batch)");
          // the group's sums: the four blocks of a row each walked a quarter of the bins
          f32x4 mv = acc + acc2;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            mv[r] += row_ror<4>(mv[r]);
            mv[r] += row_ror<8>(mv[r]);
          }
          const int nf = (info >> 8) & 7, ff = (info >> 11) & 63;
          if (floor_mel) {
            // a real branch on the wave-uniform flag (written as a select, the compare and the select
            // would run in the unfloored default too)
            MFCC_PHASE("cold: mel floor");
            asm volatile("" ::: "memory");
#pragma unroll
            for (int r = 0; r < 4; ++r) mv[r] = mv[r] == 0.f ? kEps64 : mv[r];
          }
          // ln at once (mfcc-self5.py:125-127, unfloored by default: ln 0 = -inf); the chained DCT takes
          // log2 (its weights carry the factor ln 2); filters beyond the bank's last count as 0
          float lm[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) lm[r] = OUT == kOutLogmel ? fast_ln(mv[r]) : __builtin_amdgcn_logf(mv[r]);
          if (nf < 4) {  // only a bank's last group, or a group cut in two, is short: a real branch, not four selects per group
            asm volatile("" ::: "memory");
#pragma unroll
            for (int r = 1; r < 4; ++r) lm[r] = r < nf ? lm[r] : 0.f;
          }
          if (OUT == kOutLogmel) {
            const int t = tb + frm;
            if (blk == 0 && t < s_hi) {
#pragma unroll
              for (int r = 0; r < 4; ++r)
                if (r < nf) out_t0[(t - t0) * ld + ff + r] = lm[r];
            }
          } else {
            // this group's part of the DCT-II * lifter product (mfcc-self5.py:136-156), same instruction:
            // block 4 a + b now forms coefficients 4 b .. 4 b + 3 of frames 4 a .. 4 a + 3, one k-step per filter
#pragma unroll
            for (int r = 0; r < 4; ++r) pd = __builtin_amdgcn_mfma_f32_4x4x1f32(dwc[r], lm[r], pd, 0, 0, 0);
          }
        }
      }
#else
      pd[0] = Prow[0]; pd[1] = Prow[16]; pd[2] = bv[0][0]; pd[3] = bv[1][1];
#endif
      MFCC_PHASE("9 DCT share to LDS, rows out (delta, delta-delta, stores)");
      if (OUT != kOutLogmel) *(f32x4*)__builtin_assume_aligned(xbuf + part_index(frm, wave, 4 * blk), 16) = pd;
    }
    // rows of the batch before last whose +-4-row context is now complete leave as whole output rows
    if (kRows39 && tb > s_lo) emit_rows(tb - kBatch - kHalo);
    MFCC_PHASE("10 staging: convert, pre-emphasis, LDS write");
    MFCC_STAMP(tb3);
#ifndef MFCC_ABL_STAGE
    if (more) delta = stage_st(tb + kBatch);
#endif
    MFCC_PHASE("11 barrier 2");
    MFCC_STAMP(tb4);
    __syncthreads();  // ln-mel rows and energy partials complete; next batch staged
    MFCC_PHASE("12 share sum, c0 := ln E, statics ring");
    MFCC_STAMP(tb5);
    __builtin_amdgcn_s_setprio(MFCC_PRIO_P3);

    // energy of frame slot i: the sum its transform left (4|X|^2 or 2|X| per bin), scaled like the filter weights
    auto energy_of = [&](int i) {
      float e = xbuf[i * kSlot + kPartEnergy] * spec_scale;
      if (e == 0.f) e = kEps64;  // mfcc-self5.py:130-132
      return fast_ln(e);
    };
    if (OUT == kOutLogmel) {
      // the ln-mel columns went out from registers; the energy column needs the four partials
      // (every wave for its own four frames: their slots are the ones it overwrites next)
      if (q == 0 && tb + 4 * wave + g4 < s_hi) out_t0[(tb - t0 + 4 * wave + g4) * ld + nmel] = energy_of(4 * wave + g4);
      continue;  // the tiles are rewritten only after the next spectrum barrier
    }
    {
      // ---- sum of the four DCT shares, c0 := ln E (floored at the float64 epsilon,
      // mfcc-self5.py:130-132 and :164); wave w finishes frames 4 w .. 4 w + 3, lane = (frame, coefficient)
      const int i = 4 * wave + g4;
      const int t = tb + i;
      float v = (xbuf[part_index(i, 0, q)] + xbuf[part_index(i, 1, q)]) + (xbuf[part_index(i, 2, q)] + xbuf[part_index(i, 3, q)]);
      const float e = energy_of(i);
      if (q == 0) v = e;
      if (q < ncep && t < s_hi) {
        if (kRows39) {
          const int rho = (t - s_lo + 8) & 31;
          if (kSums && t == t0) kpiv[q] = __builtin_fabsf(v) < __builtin_inff() ? v : 0.f;  // the tile's pivot: read from the next batch on
          ring[rho * RP + q] = v;
          if (kMirror && rho < 8) ring[(rho + 32) * RP + q] = v;
        } else {
          out_t0[(t - t0) * ld + q] = v;
        }
      }
    }
    // stacked MFCC-39: the rows below tb - 4 went into the ring of finished rows before the barrier above
    if (OUT == kOutMfcc39Stack && tb > s_lo) write_stack_rows(tb - kHalo);
#ifdef MFCC_DIAG
    {
      MFCC_STAMP(tb6);
      dsum[0] += tb1 - ta;   // FFT
      dsum[1] += tb2 - tb1;  // wait at spectrum barrier
      dsum[2] += tb3 - tb2;  // mel MFMA + ln
      dsum[3] += tb4 - tb3;  // staging
      dsum[4] += tb5 - tb4;  // wait at second barrier
      dsum[5] += tb6 - tb5;  // DCT
    }
#endif
  }
#ifdef MFCC_DIAG
  MFCC_STAMP(tk1);
  dsum[6] = tk1 - tk0;
#endif

  if (OUT == kOutStream) {
    // rows whose context is complete: all their static rows are in the row store now (this push's
    // written above by this workgroup, the kStreamHalo before them carried over)
    __syncthreads();
    const StreamArgs& sa = p.stream;
    const float* const S0 = p.out + (f_off + t0) * p.out_ld;  // first new static row of this stream
    float* const ob = sa.rows_out + (int64_t)blockIdx.x * sa.out_stream_stride;
    for (int idx = tid; idx < sa.k * kMaxCep; idx += 256) {
      const int c = idx & (kMaxCep - 1), i = idx / kMaxCep;
      if (c < ncep) stream_emit_one(S0 + c, sa.e0 + i, sa.lo, sa.hi, ncep, c, ob + (int64_t)i * sa.out_ld);
    }
  }
  if (kRows39) {
    // the rows still held back: from 4 rows before the last batch to the tile's end (at most 20)
    __syncthreads();
    const int tb_last = s_lo + ((s_hi - s_lo - 1) / kBatch) * kBatch;
    emit_rows(tb_last - kHalo);
    if (tb_last + kBatch - kHalo < (OUT == kOutMfcc39Stack ? o_hi : t0 + nt)) emit_rows(tb_last + kBatch - kHalo);
    if (OUT == kOutMfcc39Stack) {
      __syncthreads();
      write_stack_rows(o_hi);  // every row this tile's windows reach is in the ring now
    }
    if (kSums) {
      __syncthreads();
      if (tid < kTileSums)
        p.tile_sums[(int64_t)blockIdx.x * kTileSums + tid] =
            tid < kTileSumCols ? (csum[tid] + csum[kTileSumCols + tid]) + (csum[2 * kTileSumCols + tid] + csum[3 * kTileSumCols + tid])
                               : kpiv[tid - kTileSumCols];
    }
  }
#ifdef MFCC_DIAG
  {
    MFCC_STAMP(tk2);
    dsum[7] = tk2 - tk0;
    if (lane == 0 && p.dbg)
      for (int i = 0; i < 8; ++i) p.dbg[((size_t)blockIdx.x * 4 + wave) * 8 + i] = dsum[i];
  }
#endif
}

}  // namespace mfcc
