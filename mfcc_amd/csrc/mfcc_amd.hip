// C ABI of the MI355X MFCC front-end (see include/mfcc_amd.h).
#include "../../include/mfcc_amd.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>
#include <numeric>
#include <vector>

#include "fused_kernel.h"
#include "general_kernel.h"
#ifdef MFCC_USE_SPLIT  // experimental build: MFCC-39 through the eight-wave kernel of split_kernel.h
#include "split_kernel.h"
#endif
#include "stage_kernels.h"
#include "stream_kernels.h"
#include "tables.h"

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

#define HIP_TRY(expr)                                                                        \
  do {                                                                                       \
    hipError_t e_ = (expr);                                                                  \
    if (e_ != hipSuccess) return fail(MFCC_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
  } while (0)

template <typename T>
int upload(const std::vector<T>& host, T** dev) {
  *dev = nullptr;
  if (host.empty()) return MFCC_OK;
  HIP_TRY(hipMalloc((void**)dev, host.size() * sizeof(T)));
  HIP_TRY(hipMemcpy(*dev, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice));
  return MFCC_OK;
}

inline unsigned blocks_for(int64_t n, int per) { return (unsigned)((n + per - 1) / per); }

// Partial sums of mfcc_cmvn_accumulate: one buffer per (device, stream) that has called it, sized
// at its first call for the widest rows the entry point takes (so it is never grown, freed or
// waited for afterwards) and kept until mfcc_cmvn_release or the end of the process.  Calls on one
// stream are ordered by the stream, calls on different streams get different buffers.
constexpr int kCmvnMaxCols = 1024;
constexpr size_t kCmvnWorkspaceBytes = (size_t)mfcc::kCmvnMaxBlocks * 2 * kCmvnMaxCols * sizeof(double);  // 16 MiB
struct CmvnWorkspace {
  int device;
  hipStream_t stream;
  double* ptr;
};
std::mutex g_ws_mutex;
std::vector<CmvnWorkspace> g_ws;

int cmvn_workspace(hipStream_t stream, double** out) {
  int device = 0;
  HIP_TRY(hipGetDevice(&device));
  std::lock_guard<std::mutex> lock(g_ws_mutex);
  for (CmvnWorkspace& w : g_ws)
    if (w.device == device && w.stream == stream) {
      *out = w.ptr;
      return MFCC_OK;
    }
  double* p = nullptr;
  HIP_TRY(hipMalloc((void**)&p, kCmvnWorkspaceBytes));
  g_ws.push_back(CmvnWorkspace{device, stream, p});
  *out = p;
  return MFCC_OK;
}

// Scratch of mfcc_cmvn_accumulate_static: the same partial sums in a device array that is part of
// the code object, so the entry point never allocates.  One copy per GPU, shared by every call on
// that GPU, held until the process ends.
constexpr int kCmvnStaticCols = MFCC_CMVN_STATIC_MAX_COLS;  // >= 257 spectrum bins
constexpr size_t kCmvnStaticDoubles = (size_t)mfcc::kCmvnMaxBlocks * 2 * kCmvnStaticCols;
static __device__ __align__(256) double g_cmvn_scratch[kCmvnStaticDoubles];  // 4.3 MB
constexpr int kMaxDevices = 64;
double* g_cmvn_scratch_ptr[kMaxDevices] = {};
// Recorded behind every (non-captured) call of the static entry point on the device and waited for by
// the next one, so that calls from different streams take turns on the one array instead of racing.
hipEvent_t g_cmvn_scratch_event[kMaxDevices] = {};
std::mutex g_scratch_mutex;

// Device address of g_cmvn_scratch on the current device (resolved once per device: the array's
// name is only a host-side handle, each GPU holds its own copy at its own address).
int cmvn_static_scratch(double** out, hipEvent_t* event = nullptr) {
  int device = 0;
  HIP_TRY(hipGetDevice(&device));
  if (device < 0 || device >= kMaxDevices) return fail(MFCC_ERR_UNSUPPORTED, "device index %d outside 0..%d", device, kMaxDevices - 1);
  std::lock_guard<std::mutex> lock(g_scratch_mutex);
  if (!g_cmvn_scratch_ptr[device]) {
    void* p = nullptr;
    HIP_TRY(hipGetSymbolAddress(&p, HIP_SYMBOL(g_cmvn_scratch)));
    if (!p) return fail(MFCC_ERR_HIP, "hipGetSymbolAddress returned a null address for the CMVN scratch");
    HIP_TRY(hipEventCreateWithFlags(&g_cmvn_scratch_event[device], hipEventDisableTiming));
    g_cmvn_scratch_ptr[device] = (double*)p;
  }
  *out = g_cmvn_scratch_ptr[device];
  if (event) *event = g_cmvn_scratch_event[device];
  return MFCC_OK;
}

// Workgroups of 256 threads for the flat (16-byte piece) forms of the CMVN kernels: a multiple of
// `unit` = the smallest count for which four times the thread count is a multiple of ncol, about
// `want` of them, at most `cap`.  0 = the flat form does not apply (rows not back to back, base
// not 16-byte aligned, or no such count within the cap).
int cmvn_flat_blocks(const float* feats, int64_t feats_ld, int32_t ncol, int64_t want, int64_t cap) {
  if (feats_ld != ncol || ((uintptr_t)feats & 15)) return 0;
  const int64_t m = ncol / std::gcd((int64_t)ncol, (int64_t)4);
  const int64_t unit = m / std::gcd(m, (int64_t)256);
  if (unit > cap) return 0;
  const int64_t n = std::max<int64_t>(1, std::min(want, cap) / unit) * unit;
  return (int)n;
}

// Both stages of the CMVN reduction on `st`, partial sums in `partial` (>= nblk*2*ncol doubles).
int launch_cmvn(const float* feats, int64_t feats_ld, int64_t n_rows, int32_t ncol, double* sums, double* partial, int nblk,
                int64_t rows_per_block, hipStream_t st) {
  // rows back to back: 16-byte pieces, about 16 per thread, kCmvnMaxBlocks workgroups at most
  const int64_t n_elems = n_rows * ncol;
  const int flat = cmvn_flat_blocks(feats, feats_ld, ncol, (n_elems / 4 + 256 * 16 - 1) / (256 * 16), mfcc::kCmvnMaxBlocks);
  if (flat) {
    nblk = flat;
    hipLaunchKernelGGL(mfcc::cmvn_partial_flat_kernel, dim3((unsigned)nblk), dim3(256), 0, st, feats, n_elems, ncol, partial);
  } else {
    int threads = 256;
    while (threads < ncol) threads += 64;
    hipLaunchKernelGGL(mfcc::cmvn_partial_kernel, dim3((unsigned)nblk), dim3(threads), 2 * threads * sizeof(double), st, feats, feats_ld,
                       n_rows, ncol, rows_per_block, partial);
  }
  hipLaunchKernelGGL(mfcc::cmvn_finish_kernel, dim3(2 * (unsigned)ncol), dim3(256), 0, st, partial, nblk, ncol, n_rows, sums);
  HIP_TRY(hipGetLastError());
  return MFCC_OK;
}

// Rows per stage-1 workgroup: 512 while that needs at most kCmvnMaxBlocks workgroups, more beyond.
void cmvn_split(int64_t n_rows, int* nblk, int64_t* rows_per_block) {
  int64_t rpb = 512;
  if ((n_rows + rpb - 1) / rpb > mfcc::kCmvnMaxBlocks) rpb = (n_rows + mfcc::kCmvnMaxBlocks - 1) / mfcc::kCmvnMaxBlocks;
  *rows_per_block = rpb;
  *nblk = (int)((n_rows + rpb - 1) / rpb);
}

}  // namespace

struct mfcc_plan {
  mfcc_config cfg;
  int device;
  int32_t frame_size, frame_step, nfft;
  float* tw512 = nullptr;
  mfcc::PlanConst pc_host;
  mfcc::PlanConst* pc = nullptr;
  float* melW = nullptr;
  float* dctW = nullptr;
  float* window = nullptr;
  float pre_hi, pre_lo, spec_scale;
  // general kernel (general_kernel.h): set when the configuration is outside the specialised kernel
  bool general = false;
  // 320/160-sample frames (frame_time = 20 at 16 kHz): the specialised kernel, instantiated for the kinds of
  // mfcc_forward only -- stacked rows take mfcc_forward + mfcc_stack_context, streams the three-launch form
  bool plain_only = false;
  int32_t gen_nt_max = 0;  // output frames per tile
  bool gen_no_delta = false;  // no room in LDS for a tile's static / delta rows: every kind but MFCC-39
  size_t gen_smem = 0;     // dynamic LDS per workgroup
  mfcc::GeneralConst gc_host;
  mfcc::GeneralConst* gc = nullptr;
  float* g_tw = nullptr;
  float* g_bank = nullptr;
  int32_t* g_band = nullptr;
  float* g_dctl = nullptr;
  float* g_window = nullptr;
};

struct mfcc_layout {
  int device;
  int64_t n_utts, total_samples, total_frames;
  std::vector<int64_t> sample_off, frame_off;
  int64_t* d_sample_off = nullptr;
  int64_t* d_frame_off = nullptr;
  mfcc::Tile* d_tiles = nullptr;
  int64_t n_tiles;
  // the same frames cut for the persistent workgroups of split_kernel.h: workgroup w walks
  // d_wg_tiles[d_wg_off[w] .. d_wg_off[w + 1]); every workgroup gets about the same number of batches
  mfcc::Tile* d_wg_tiles = nullptr;
  int32_t* d_wg_off = nullptr;
  int32_t n_wg = 0;
  bool frames_only = false;  // made by mfcc_layout_create_frames: frame offsets only, no audio geometry
  // what the tile list was cut for: a plan may only run on a layout of its own framing whose
  // tiles fit its kernel
  int32_t frame_size = 0, frame_step = 0, frame_rule = 0;
  int32_t max_nt = 0;  // most output frames in one tile
  bool general_cut = false;  // tiles cut for the general kernel (no halo rule)
};

struct mfcc_stream {
  const mfcc_plan* plan = nullptr;
  int device = 0;
  int32_t n_streams = 0, max_chunk = 0, dtype = 0, cap = 0, rows = 0, tile_max = 0, max_tiles = 0;
  void* seg = nullptr;             // [2][n_streams][cap] samples
  float* stat = nullptr;           // [2][n_streams][rows][kMaxCep]
  int64_t* d_sample_off = nullptr; // per-step layout of the fused kernel, written on the device
  int64_t* d_sample_len = nullptr;
  int64_t* d_frame_off = nullptr;
  mfcc::Tile* d_tiles = nullptr;
  // host-side state, identical for all streams (they advance in lock step)
  int64_t total = 0;      // samples pushed since the utterance began
  int64_t seg_start = 0;  // utterance sample held at element 0 of the current segment
  int64_t f_done = 0;     // static rows computed
  int64_t e_done = 0;     // output rows emitted
  int32_t half = 0, prev_new_rows = 0;
};

#ifdef MFCC_DIAG
static unsigned long long* g_dbg = nullptr;
extern "C" void mfcc_debug_set_buffer(void* p) { g_dbg = (unsigned long long*)p; }
#endif

#ifndef MFCC_EXTRA_SMEM
#define MFCC_EXTRA_SMEM 0  // occupancy experiments only
#endif

// launch_fused selectors of the stacked spectrum kinds (not part of the public out_kind values)
enum : int32_t { kLaunchSpecStack = 100, kLaunchLogspecStack = 101, kLaunchStream = 102, kLaunchMfcc39Stack = 103, kLaunchMfcc39Sums = 104 };

// The stacked-spectrum kernels use more dynamic LDS than a kernel may have without asking.  Asked once
// per device, at plan creation (with that device current), so that a launch is nothing but a launch
// (it may sit inside a stream capture).
static int allow_stacked_lds() {
  const int bytes = (int)(mfcc::smem_floats_stacked(mfcc::kStackMaxN) * sizeof(float)) + MFCC_EXTRA_SMEM;
#define MFCC_ALLOW(OUT, T, G) \
  HIP_TRY(hipFuncSetAttribute((const void*)mfcc::fused_kernel<OUT, 0, T, G>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes))
  MFCC_ALLOW(mfcc::kOutSpecStack, int16_t, mfcc::Geom16k);
  MFCC_ALLOW(mfcc::kOutSpecStack, float, mfcc::Geom16k);
  MFCC_ALLOW(mfcc::kOutSpecStack, int16_t, mfcc::Geom8k);
  MFCC_ALLOW(mfcc::kOutSpecStack, float, mfcc::Geom8k);
  MFCC_ALLOW(mfcc::kOutLogspecStack, int16_t, mfcc::Geom16k);
  MFCC_ALLOW(mfcc::kOutLogspecStack, float, mfcc::Geom16k);
  MFCC_ALLOW(mfcc::kOutLogspecStack, int16_t, mfcc::Geom8k);
  MFCC_ALLOW(mfcc::kOutLogspecStack, float, mfcc::Geom8k);
#undef MFCC_ALLOW
  // the stacked MFCC-39 kernels: statics ring + ring of finished rows
  const int bytes39 = (int)((mfcc::smem_floats(mfcc::kMaxCep) + mfcc::oring_floats(mfcc::kMaxCep)) * sizeof(float)) + MFCC_EXTRA_SMEM;
#define MFCC_ALLOW39(NCEP, T, G) \
  HIP_TRY(hipFuncSetAttribute((const void*)mfcc::fused_kernel<mfcc::kOutMfcc39Stack, NCEP, T, G>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes39))
  MFCC_ALLOW39(13, int16_t, mfcc::Geom16k);
  MFCC_ALLOW39(13, float, mfcc::Geom16k);
  MFCC_ALLOW39(13, int16_t, mfcc::Geom8k);
  MFCC_ALLOW39(13, float, mfcc::Geom8k);
  MFCC_ALLOW39(0, int16_t, mfcc::Geom16k);
  MFCC_ALLOW39(0, float, mfcc::Geom16k);
  MFCC_ALLOW39(0, int16_t, mfcc::Geom8k);
  MFCC_ALLOW39(0, float, mfcc::Geom8k);
#undef MFCC_ALLOW39
#ifdef MFCC_USE_SPLIT
  const int bytes_split = (int)(mfcc::split_smem_floats(mfcc::kMaxCep) * sizeof(float));
#define MFCC_ALLOW_SPLIT(NCEP, T, G) \
  HIP_TRY(hipFuncSetAttribute((const void*)mfcc::split_kernel<NCEP, T, G>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes_split))
  MFCC_ALLOW_SPLIT(13, int16_t, mfcc::Geom16k);
  MFCC_ALLOW_SPLIT(13, float, mfcc::Geom16k);
  MFCC_ALLOW_SPLIT(13, int16_t, mfcc::Geom8k);
  MFCC_ALLOW_SPLIT(13, float, mfcc::Geom8k);
  MFCC_ALLOW_SPLIT(0, int16_t, mfcc::Geom16k);
  MFCC_ALLOW_SPLIT(0, float, mfcc::Geom16k);
  MFCC_ALLOW_SPLIT(0, int16_t, mfcc::Geom8k);
  MFCC_ALLOW_SPLIT(0, float, mfcc::Geom8k);
#undef MFCC_ALLOW_SPLIT
#endif
  return MFCC_OK;
}

// One workgroup per tile; template arguments select the output kind, a compile-time numcep of 13
// where it pays, and the sample type.
static void launch_fused(const mfcc_plan* plan, const mfcc::FusedParams& fp, int64_t n_tiles, int32_t out_kind, bool i16,
                         hipStream_t st, int32_t n_wg = 0) {
  const dim3 grid((unsigned)n_tiles), block(256);
  if (plan->general) {
    // the public out_kind values are the kernel's own
    const bool lds_tables = plan->gc_host.bankc_lds > 0 && plan->gc_host.dct_lds > 0;
    if (plan->gc_host.lane16) {
      if (lds_tables) {
        if (i16) hipLaunchKernelGGL((mfcc::general16_kernel<int16_t, true>), grid, block, plan->gen_smem, st, fp, plan->gc, (int)out_kind);
        else hipLaunchKernelGGL((mfcc::general16_kernel<float, true>), grid, block, plan->gen_smem, st, fp, plan->gc, (int)out_kind);
      } else {
        if (i16) hipLaunchKernelGGL((mfcc::general16_kernel<int16_t, false>), grid, block, plan->gen_smem, st, fp, plan->gc, (int)out_kind);
        else hipLaunchKernelGGL((mfcc::general16_kernel<float, false>), grid, block, plan->gen_smem, st, fp, plan->gc, (int)out_kind);
      }
    } else {
      // 1024 points: 32 lanes per frame (R = 2); 2048 points: 64 lanes per frame (R = 4)
#define MFCC_LAUNCH_W(R)                                                                                                                      \
  do {                                                                                                                                        \
    if (lds_tables) {                                                                                                                         \
      if (i16) hipLaunchKernelGGL((mfcc::generalw_kernel<int16_t, true, R>), grid, block, plan->gen_smem, st, fp, plan->gc, (int)out_kind);   \
      else hipLaunchKernelGGL((mfcc::generalw_kernel<float, true, R>), grid, block, plan->gen_smem, st, fp, plan->gc, (int)out_kind);         \
    } else {                                                                                                                                  \
      if (i16) hipLaunchKernelGGL((mfcc::generalw_kernel<int16_t, false, R>), grid, block, plan->gen_smem, st, fp, plan->gc, (int)out_kind);  \
      else hipLaunchKernelGGL((mfcc::generalw_kernel<float, false, R>), grid, block, plan->gen_smem, st, fp, plan->gc, (int)out_kind);        \
    }                                                                                                                                         \
  } while (0)
      if (plan->nfft == 1024) MFCC_LAUNCH_W(2);
      else MFCC_LAUNCH_W(4);
#undef MFCC_LAUNCH_W
    }
    return;
  }
  const bool c13 = plan->cfg.numcep == 13;
  // only the MFCC-39 kind keeps a ring of static rows in LDS: 13 floats per row where numcep is a compile-time 13
  const bool rows39 = out_kind == MFCC_OUT_MFCC39 || out_kind == kLaunchMfcc39Stack || out_kind == kLaunchMfcc39Sums;
  const int ring_pitch = rows39 ? (c13 ? 13 : mfcc::kMaxCep) : 0;
  const bool stacked = out_kind == kLaunchSpecStack || out_kind == kLaunchLogspecStack;
  const size_t smem = (stacked ? mfcc::smem_floats_stacked(fp.stack_n)
                               : mfcc::smem_floats(ring_pitch) + (out_kind == kLaunchMfcc39Stack ? mfcc::oring_floats(plan->cfg.numcep) : 0) +
                                     (out_kind == kLaunchMfcc39Sums ? mfcc::kSumsLdsFloats : 0)) *
                          sizeof(float) + MFCC_EXTRA_SMEM;
  const bool g8k = plan->frame_size == mfcc::Geom8k::kSize, g20 = plan->frame_size == mfcc::Geom20ms::kSize;
  // (the stacked kinds take more than the 64 KiB of dynamic LDS a kernel gets without asking: allow_stacked_lds())
#define MFCC_LAUNCH_G(OUT, NCEP, G)                                                                          \
  do {                                                                                                       \
    if (i16) hipLaunchKernelGGL((mfcc::fused_kernel<OUT, NCEP, int16_t, G>), grid, block, smem, st, fp);     \
    else hipLaunchKernelGGL((mfcc::fused_kernel<OUT, NCEP, float, G>), grid, block, smem, st, fp);           \
  } while (0)
#define MFCC_LAUNCH(OUT, NCEP)                                 \
  do {                                                         \
    if (g8k) MFCC_LAUNCH_G(OUT, NCEP, mfcc::Geom8k);           \
    else MFCC_LAUNCH_G(OUT, NCEP, mfcc::Geom16k);              \
  } while (0)
  // the kinds of mfcc_forward also exist for 320/160-sample frames
#define MFCC_LAUNCH3(OUT, NCEP)                                \
  do {                                                         \
    if (g20) MFCC_LAUNCH_G(OUT, NCEP, mfcc::Geom20ms);         \
    else MFCC_LAUNCH(OUT, NCEP);                               \
  } while (0)
#ifdef MFCC_USE_SPLIT
  if (out_kind == MFCC_OUT_MFCC39 && n_wg > 0 && fp.wg_tiles && !g20) {
    // eight waves per workgroup in two roles, two persistent workgroups per CU (split_kernel.h)
    const size_t smem_s = mfcc::split_smem_floats(ring_pitch) * sizeof(float);
    const dim3 block8(512);
    const dim3 grid((unsigned)n_wg);
#define MFCC_LAUNCH_S(NCEP, G)                                                                                 \
  do {                                                                                                         \
    if (i16) hipLaunchKernelGGL((mfcc::split_kernel<NCEP, int16_t, G>), grid, block8, smem_s, st, fp);         \
    else hipLaunchKernelGGL((mfcc::split_kernel<NCEP, float, G>), grid, block8, smem_s, st, fp);               \
  } while (0)
    if (c13) { if (g8k) MFCC_LAUNCH_S(13, mfcc::Geom8k); else MFCC_LAUNCH_S(13, mfcc::Geom16k); }
    else { if (g8k) MFCC_LAUNCH_S(0, mfcc::Geom8k); else MFCC_LAUNCH_S(0, mfcc::Geom16k); }
#undef MFCC_LAUNCH_S
    return;
  }
#endif
  switch (out_kind) {
    case MFCC_OUT_MFCC39:
      if (c13) MFCC_LAUNCH3(mfcc::kOutMfcc39, 13);
      else MFCC_LAUNCH3(mfcc::kOutMfcc39, 0);
      break;
    case MFCC_OUT_STATIC:
      if (c13) MFCC_LAUNCH3(mfcc::kOutStatic, 13);
      else MFCC_LAUNCH3(mfcc::kOutStatic, 0);
      break;
    case MFCC_OUT_LOGMEL: MFCC_LAUNCH3(mfcc::kOutLogmel, 0); break;
    case MFCC_OUT_SPEC: MFCC_LAUNCH3(mfcc::kOutSpec, 0); break;
    case MFCC_OUT_LOGSPEC: MFCC_LAUNCH3(mfcc::kOutLogspec, 0); break;
    case kLaunchSpecStack: MFCC_LAUNCH(mfcc::kOutSpecStack, 0); break;
    case kLaunchLogspecStack: MFCC_LAUNCH(mfcc::kOutLogspecStack, 0); break;
    case kLaunchStream:
      if (c13) MFCC_LAUNCH(mfcc::kOutStream, 13);
      else MFCC_LAUNCH(mfcc::kOutStream, 0);
      break;
    case kLaunchMfcc39Stack:
      if (c13) MFCC_LAUNCH(mfcc::kOutMfcc39Stack, 13);
      else MFCC_LAUNCH(mfcc::kOutMfcc39Stack, 0);
      break;
    case kLaunchMfcc39Sums:
      if (c13) MFCC_LAUNCH3(mfcc::kOutMfcc39Sums, 13);
      else MFCC_LAUNCH3(mfcc::kOutMfcc39Sums, 0);
      break;
  }
#undef MFCC_LAUNCH_G
#undef MFCC_LAUNCH
#undef MFCC_LAUNCH3
}

// A layout's tile list was cut for the plan it was created with; another plan may use it when
// the framing is the same and every tile fits that plan's kernel.
static int layout_fits(const mfcc_plan* plan, const mfcc_layout* layout) {
  if (layout->frame_size != plan->frame_size || layout->frame_step != plan->frame_step || layout->frame_rule != plan->cfg.frame_rule)
    return fail(MFCC_ERR_INVALID, "the layout was created for frames of %d samples every %d (frame_rule %d), the plan has %d every %d (frame_rule %d)",
                layout->frame_size, layout->frame_step, layout->frame_rule, plan->frame_size, plan->frame_step, plan->cfg.frame_rule);
  // the specialised kernel takes the tiles its own rule cut (up to a whole 19-batch tile when the
  // utterance ends inside it) and any tile that leaves room for a halo on both sides
  const int cap = plan->general ? plan->gen_nt_max : mfcc::tile_frames_for(plan->cfg.numcep) - 2 * mfcc::kHalo;
  if (layout->max_nt > cap && (plan->general || layout->general_cut))
    return fail(MFCC_ERR_INVALID, "the layout's tiles hold up to %d frames, this plan's kernel takes %d: create the layout with this plan",
                layout->max_nt, cap);
  return MFCC_OK;
}

extern "C" {

int mfcc_version(void) { return 101; }

const char* mfcc_last_error(void) { return g_err; }

int32_t mfcc_config_size(void) { return (int32_t)sizeof(mfcc_config); }

void mfcc_config_default(mfcc_config* cfg) {
  if (!cfg) return;
  cfg->fs = 16000;
  cfg->frame_time = 25;
  cfg->step_time = 10;
  cfg->num_bins = 26;
  cfg->low_freq = 0;
  cfg->numcep = 13;
  cfg->lift_factor = 22;
  cfg->pad_fac = 2;
  cfg->use_preemph = 1;
  cfg->preemph = 0.97;
  cfg->spectrum = MFCC_SPECTRUM_POWER;
  cfg->window = MFCC_WINDOW_NONE;
  cfg->floor_mel = 0;
  cfg->num_fft = 0;
  cfg->frame_rule = 0;
  cfg->log_floor = 0.0;
  cfg->high_freq = 0.0;
  cfg->kernel = MFCC_KERNEL_AUTO;
  cfg->keep_c0 = 0;
}

int mfcc_melbank_host(double fs, int32_t numbins, double lowfreq, int32_t num_fft, double* out_host) {
  if (!out_host || numbins < 1 || num_fft < 2 || fs <= 0) return fail(MFCC_ERR_INVALID, "mfcc_melbank_host: bad argument");
  const std::vector<double> b = mfcc::melbank(fs, numbins, lowfreq, num_fft);
  memcpy(out_host, b.data(), b.size() * sizeof(double));
  return MFCC_OK;
}

int mfcc_frame_geometry(double fs, double frame_time, double step_time, int32_t* frame_size, int32_t* frame_step,
                        int32_t* num_fft) {
  if (!frame_size || !frame_step || !num_fft) return fail(MFCC_ERR_INVALID, "mfcc_frame_geometry: null output");
  mfcc::frame_geometry(fs, frame_time, step_time, frame_size, frame_step, num_fft);
  return MFCC_OK;
}

int64_t mfcc_frames_for_length(int64_t n_samples, int32_t frame_size, int32_t frame_step) {
  if (n_samples < 0 || frame_size < 1 || frame_step < 1) return fail(MFCC_ERR_INVALID, "mfcc_frames_for_length: bad argument");
  return mfcc::frames_for_length(n_samples, frame_size, frame_step);
}

// Tables of the specialised kernel.  Returns 1 (and leaves the plan without them) when the
// filterbank does not fit that kernel's slots of matrix-core k-steps per wave (a bank of few, very
// wide filters: a group of four then covers more bins than a wave walks).
static int build_specialised(mfcc_plan* p) {
  const mfcc_config* const cfg = &p->cfg;
  const int32_t nfft = p->nfft, size = p->frame_size;
  // ---- operands of the two matrix-core products (layout: see mfcc::MelMap) ----------------
  const int nmel = cfg->num_bins;
  // The kernel always transforms 512 points.  A 256-point FFT of a frame of at most 256 samples
  // (8 kHz: 200 samples, mfcc-self5.py:79-84) is every other bin of the 512-point FFT of the same
  // zero-padded frame, X256[j] = X512[2 j]: the filterbank and the energy sum of such a plan carry
  // their weights on the even bins of the 257-bin row and zeros on the odd ones.
  const int decim = mfcc::kNfft / nfft;  // 1 or 2
  std::vector<double> bank((size_t)nmel * mfcc::kNbin, 0.0);
  {
    const std::vector<double> own = mfcc::melbank(cfg->fs, nmel, cfg->low_freq, nfft, cfg->high_freq);
    const int nb = nfft / 2 + 1;
    for (int m = 0; m < nmel; ++m)
      for (int k = 0; k < nb; ++k) bank[(size_t)m * mfcc::kNbin + (size_t)k * decim] = own[(size_t)m * nb + k];
  }
  // Groups of up to four consecutive filters; each walks the bins that carry a weight of one of its filters,
  // from a multiple of four on, in quads of 16 bins that end inside the zero-padded spectrum row.
  struct Group { int first, nfilt, k0, quads; };
  auto measure = [&](Group& gr) {
    int lo = mfcc::kNbin, hi = 0;
    for (int m = gr.first; m < gr.first + gr.nfilt; ++m)
      for (int k = 0; k < mfcc::kNbin; ++k)
        if (bank[(size_t)m * mfcc::kNbin + k] != 0.0) { lo = std::min(lo, k); hi = std::max(hi, k + 1); }
    if (hi <= lo) { lo = 0; hi = 1; }  // a group without any weight (degenerate bank) still yields ln 0 for its filters
    gr.k0 = lo & ~3;
    gr.quads = (hi - gr.k0 + 15) / 16;
    if (gr.k0 + 16 * gr.quads > mfcc::kRowPad) gr.k0 = mfcc::kRowPad - 16 * gr.quads;
    static_assert(mfcc::kRowPad % 4 == 0, "clamped start stays 16-byte aligned");
  };
  std::vector<Group> groups;
  for (int f = 0; f < nmel; f += 4) {
    groups.push_back(Group{f, std::min(4, nmel - f), 0, 0});
    measure(groups.back());
  }
  // Longest group first onto the wave with the least work so far (cost in matrix instructions: four per
  // quad, four for the group's part of the DCT product); a wave takes at most kMelQuads quads.  While that
  // fails (few, wide filters), the widest group that still has two filters is cut in two.
  std::vector<Group> mine[4];
  for (;;) {
    std::vector<Group> order = groups;
    std::stable_sort(order.begin(), order.end(), [](const Group& x, const Group& y) { return x.quads > y.quads; });
    int load[4] = {0, 0, 0, 0}, used[4] = {0, 0, 0, 0};
    for (int w = 0; w < 4; ++w) mine[w].clear();
    bool ok = (int)groups.size() <= mfcc::kMelGroups;
    for (size_t i = 0; i < order.size() && ok; ++i) {
      const Group& gr = order[i];
      int best = -1;
      for (int w = 0; w < 4; ++w)
        if (used[w] + gr.quads <= mfcc::kMelQuads && (best < 0 || load[w] < load[best])) best = w;
      if (best < 0) { ok = false; break; }
      mine[best].push_back(gr);
      load[best] += 4 * gr.quads + 4;
      used[best] += gr.quads;
    }
    if (ok) break;
    int widest = -1;
    for (size_t i = 0; i < groups.size(); ++i)
      if (groups[i].nfilt > 1 && (widest < 0 || groups[i].quads > groups[(size_t)widest].quads)) widest = (int)i;
    if (widest < 0 || (int)groups.size() >= mfcc::kMelGroups) return 1;  // the bank does not fit the specialised kernel's slots
    Group lo_half = groups[(size_t)widest], hi_half = lo_half;
    lo_half.nfilt = (lo_half.nfilt + 1) / 2;
    hi_half.first += lo_half.nfilt;
    hi_half.nfilt -= lo_half.nfilt;
    measure(lo_half);
    measure(hi_half);
    groups[(size_t)widest] = lo_half;
    groups.insert(groups.begin() + widest + 1, hi_half);
  }
  const int ngroups = (int)groups.size();
  auto group_number = [&](const Group& gr) {
    for (int g = 0; g < ngroups; ++g)
      if (groups[(size_t)g].first == gr.first) return g;
    return 0;
  };
  mfcc::PlanConst& pch = p->pc_host;
  memset(&pch, 0, sizeof(pch));
  mfcc::MelMap& pmap = pch.map;
  std::vector<float> melW((size_t)4 * mfcc::kMelQuads * 16 * 4, 0.f);
  for (int w = 0; w < 4; ++w) {
    int slot = 0;
    for (const Group& gr : mine[w])
      for (int i = 0; i < gr.quads; ++i, ++slot) {
        const int bin0 = gr.k0 + 16 * i;
        pmap.slot[w][slot] = (uint32_t)(bin0 / 4) | (i == 0 ? mfcc::kSlotFirst : 0u) | (i == gr.quads - 1 ? mfcc::kSlotLast : 0u) |
                             ((uint32_t)gr.nfilt << 8) | ((uint32_t)gr.first << 11) | ((uint32_t)group_number(gr) << 17);
        for (int b = 0; b < 4; ++b)
          for (int f = 0; f < gr.nfilt; ++f)
            for (int st = 0; st < 4; ++st) {
              const int k = bin0 + 4 * b + st;
              const double wgt = k < mfcc::kNbin ? bank[(size_t)(gr.first + f) * mfcc::kNbin + k] : 0.0;
              melW[(((size_t)w * mfcc::kMelQuads + slot) * 16 + 4 * b + f) * 4 + st] = (float)(wgt * (double)p->spec_scale);
            }
      }
    pmap.nslots[w] = slot;
  }
  const std::vector<double> dl = mfcc::dct_lifter(nmel, cfg->numcep, cfg->lift_factor, true);
  // weights of the chained product (see PlanConst::dctW); the kernel hands over log2 of the mel sums
  std::vector<float> dctW((size_t)mfcc::kMelGroups * 64, 0.f);
  for (int g = 0; g < ngroups; ++g)
    for (int b = 0; b < 4; ++b)
      for (int i = 0; i < 4; ++i)
        for (int r = 0; r < groups[(size_t)g].nfilt; ++r) {
          const int c = 4 * b + i, m = groups[(size_t)g].first + r;
          if (c < cfg->numcep && m < nmel) dctW[(size_t)g * 64 + (4 * b + i) * 4 + r] = (float)(dl[(size_t)m * cfg->numcep + c] * M_LN2);
        }
  std::vector<float> win;
  if (cfg->window == MFCC_WINDOW_HAMMING) {
    const std::vector<double> h = mfcc::hamming(size);
    win.assign(512, 0.f);
    for (int n = 0; n < size; ++n) win[n] = (float)h[n];
  }
  int rc;
  if ((rc = upload(mfcc::twiddle512(), &p->tw512)) ||
      (rc = upload(melW, &p->melW)) || (rc = upload(dctW, &p->dctW)) || (rc = upload(win, &p->window))) {
    return rc;
  }
  pch.tw512 = p->tw512;
  pch.melW = p->melW;
  pch.dctW = p->dctW;
  pch.window = p->window;
  pch.nmel = nmel;
  pch.numcep = cfg->numcep;
  pch.pre_hi = cfg->use_preemph ? p->pre_hi : 0.f;
  pch.pre_lo = cfg->use_preemph ? p->pre_lo : 0.f;
  pch.spectrum = cfg->spectrum;
  pch.floor_mel = cfg->floor_mel;
  pch.spec_scale = p->spec_scale;
  pch.log_floor = (float)cfg->log_floor;
  pch.decim = decim;
  if ((rc = upload(std::vector<mfcc::PlanConst>(1, pch), &p->pc))) {
    return rc;
  }
  return MFCC_OK;
}

// Tables of the general kernel (see general_kernel.h).
static int build_general(mfcc_plan* p) {
  const mfcc_config* const cfg = &p->cfg;
  const int nfft = p->nfft, nbin = nfft / 2 + 1, nmel = cfg->num_bins, ncep = cfg->numcep, pf = cfg->pad_fac;
  // filter weights, compact: filter m's weights of the bins band[3m] .. band[3m+1]-1 start at band[3m+2]
  const std::vector<double> bank = mfcc::melbank(cfg->fs, nmel, cfg->low_freq, nfft, cfg->high_freq);
  std::vector<float> bankc;
  std::vector<int32_t> band((size_t)3 * nmel, 0);
  for (int m = 0; m < nmel; ++m) {
    int lo = nbin, hi = 0;
    for (int k = 0; k < nbin; ++k)
      if (bank[(size_t)m * nbin + k] != 0.0) { lo = std::min(lo, k); hi = std::max(hi, k + 1); }
    lo = std::min(lo, hi);
    band[3 * m] = lo;
    band[3 * m + 1] = hi;
    band[3 * m + 2] = (int32_t)bankc.size();
    for (int k = lo; k < hi; ++k) bankc.push_back((float)bank[(size_t)m * nbin + k]);
  }
  if (bankc.empty()) bankc.push_back(0.f);
  // LDS budget: 64 KiB per workgroup.  The twiddles and the four waves' buffers always live there;
  // the filter weights and the DCT matrix join them when the tile's static and delta rows still
  // get a useful share of the rest (they are read from global memory otherwise).
  const size_t fixed_bytes = mfcc::general_fixed_floats(nfft, nmel) * sizeof(float);
  const size_t row_bytes = (size_t)2 * ncep * sizeof(float);
  const size_t want_rows = (size_t)4 * pf + 32;
  size_t table_floats = 0;
  int bank_lds = 0, dct_lds = 0;
  const size_t bank_pad = (bankc.size() + 3) & ~(size_t)3, dct_pad = ((size_t)nmel * ncep + 3) & ~(size_t)3;
  bankc.resize(bank_pad, 0.f);  // the kernel copies whole 16-byte pieces
  // both or neither: the kernel is compiled once for tables in LDS and once for tables in global memory
  if (fixed_bytes + (bank_pad + dct_pad) * sizeof(float) + want_rows * row_bytes <= 65536) {
    bank_lds = dct_lds = 1;
    table_floats = bank_pad + dct_pad;
  }
  const int64_t rows_fit = ((int64_t)65536 - (int64_t)fixed_bytes - (int64_t)(table_floats * sizeof(float))) / (int64_t)row_bytes;
  int rows = (int)std::min<int64_t>(rows_fit, mfcc::kGenMaxTile + 4 * pf);
  {
    // The general kernels wait on memory and LDS round trips more than on arithmetic: a third
    // workgroup per CU (LDS <= 42 of the CU's 128 allocation units) is worth more than tiles of 128 frames,
    // as long as a tile still holds 48.
    const int64_t rows3 = ((int64_t)42 * 1280 - (int64_t)fixed_bytes - (int64_t)(table_floats * sizeof(float))) / (int64_t)row_bytes;
    if (rows3 - 4 * pf >= 48) rows = (int)std::min<int64_t>(rows, rows3);
  }
  if (rows - 4 * pf < 4) {
    // Too little LDS left for the static and delta rows of even a four-frame tile (e.g. 2048-point FFT,
    // 128 filters, 64 cepstra): only the MFCC-39 kind uses those rows, so the plan serves every other
    // kind and mfcc_forward refuses that one.
    if ((int64_t)fixed_bytes + (int64_t)(table_floats * sizeof(float)) > 65536)
      return fail(MFCC_ERR_UNSUPPORTED, "num_fft=%d, num_bins=%d need more LDS than a workgroup has", nfft, nmel);
    p->gen_no_delta = true;
    rows = 0;
  }
  p->gen_nt_max = p->gen_no_delta ? mfcc::kGenMaxTile : rows - 4 * pf;
  p->gen_smem = mfcc::general_smem_bytes(nfft, nmel, ncep, rows) + table_floats * sizeof(float);

  const std::vector<double> dl = mfcc::dct_lifter(nmel, ncep, cfg->lift_factor, true);
  std::vector<float> dlf(dl.begin(), dl.end());
  std::vector<float> win;
  if (cfg->window == MFCC_WINDOW_HAMMING) {
    const std::vector<double> h = mfcc::hamming(p->frame_size);
    win.assign(h.begin(), h.end());
  }
  int rc;
  const bool lane16 = nfft <= mfcc::kNfft;  // the four-frames-per-wave kernel on the 512-point register transform
  if ((rc = upload(lane16 ? mfcc::twiddle512() : mfcc::twiddle_half(nfft), &p->g_tw)) || (rc = upload(bankc, &p->g_bank)) || (rc = upload(band, &p->g_band)) ||
      (rc = upload(dlf, &p->g_dctl)) || (rc = upload(win, &p->g_window)))
    return rc;
  mfcc::GeneralConst& g = p->gc_host;
  memset(&g, 0, sizeof(g));
  g.tw = p->g_tw;
  g.bankc = p->g_bank;
  g.bankc_lds = bank_lds ? (int32_t)bank_pad : 0;
  g.dct_lds = dct_lds ? (int32_t)dct_pad : 0;
  g.band = p->g_band;
  g.dctl = p->g_dctl;
  g.window = p->g_window;
  g.frame_size = p->frame_size;
  g.frame_step = p->frame_step;
  g.nfft = nfft;
  g.log2m = 0;
  while ((2 << g.log2m) < nfft) ++g.log2m;
  g.used = std::min(p->frame_size, nfft);
  g.nmel = nmel;
  g.numcep = ncep;
  g.pad_fac = pf;
  g.spectrum = cfg->spectrum;
  g.floor_mel = cfg->floor_mel;
  g.pre_hi = cfg->use_preemph ? p->pre_hi : 0.f;
  g.pre_lo = cfg->use_preemph ? p->pre_lo : 0.f;
  g.inv_nfft = 1.0f / (float)nfft;
  g.log_floor = (float)cfg->log_floor;
  int denom = 0;
  for (int i = 1; i <= pf; ++i) denom += 2 * i * i;  // mfcc-self5.py:172
  g.delta_denom = (float)denom;
  g.keep_c0 = cfg->keep_c0 ? 1 : 0;
  g.stat_rows = rows;
  g.lane16 = lane16 ? 1 : 0;
  if ((rc = upload(std::vector<mfcc::GeneralConst>(1, g), &p->gc))) return rc;
  p->general = true;
  return MFCC_OK;
}

int mfcc_plan_create(const mfcc_config* cfg, int device, mfcc_plan** out) {
  if (!cfg || !out) return fail(MFCC_ERR_INVALID, "mfcc_plan_create: null argument");
  *out = nullptr;
  if (!(cfg->fs > 0) || !(cfg->frame_time > 0) || !(cfg->step_time > 0))
    return fail(MFCC_ERR_INVALID, "fs=%g, frame_time=%g, step_time=%g must be positive", cfg->fs, cfg->frame_time, cfg->step_time);
  if (cfg->fs * cfg->frame_time > 1e9 || cfg->fs * cfg->step_time > 1e9)
    return fail(MFCC_ERR_UNSUPPORTED, "fs=%g, frame_time=%g, step_time=%g: frame far beyond %d samples", cfg->fs, cfg->frame_time,
                cfg->step_time, mfcc::kGenMaxFft);
  int32_t size, step, nfft;
  mfcc::frame_geometry(cfg->fs, cfg->frame_time, cfg->step_time, &size, &step, &nfft);
  if (size < 1 || step < 1)
    return fail(MFCC_ERR_INVALID, "fs=%g with %g ms frames every %g ms gives frames of %d samples every %d: both must be >= 1", cfg->fs,
                cfg->frame_time, cfg->step_time, size, step);
  if (cfg->num_fft != 0) nfft = cfg->num_fft;
  if (nfft < 2 || nfft > mfcc::kGenMaxFft || (nfft & (nfft - 1)))
    return fail(MFCC_ERR_UNSUPPORTED, "framing %d/%d samples with a %d-point FFT: the kernels take FFT lengths 2, 4, ... %d "
                "(frame sizes up to %d samples when num_fft is derived as winfft does)", size, step, nfft, mfcc::kGenMaxFft, mfcc::kGenMaxFft);
  if (cfg->num_bins < 1 || cfg->num_bins > mfcc::kGenMaxMel)
    return fail(MFCC_ERR_UNSUPPORTED, "num_bins=%d outside 1..%d", cfg->num_bins, mfcc::kGenMaxMel);
  if (cfg->numcep < 1 || cfg->numcep > mfcc::kGenMaxCep || cfg->numcep > cfg->num_bins)
    return fail(MFCC_ERR_UNSUPPORTED, "numcep=%d outside 1..min(%d, num_bins)", cfg->numcep, mfcc::kGenMaxCep);
  if (cfg->pad_fac < 1 || cfg->pad_fac > mfcc::kGenMaxPad)
    return fail(MFCC_ERR_UNSUPPORTED, "pad_fac=%d outside 1..%d", cfg->pad_fac, mfcc::kGenMaxPad);
  if (cfg->spectrum != MFCC_SPECTRUM_POWER && cfg->spectrum != MFCC_SPECTRUM_MAG)
    return fail(MFCC_ERR_INVALID, "spectrum=%d", cfg->spectrum);
  if (cfg->window != MFCC_WINDOW_NONE && cfg->window != MFCC_WINDOW_HAMMING)
    return fail(MFCC_ERR_INVALID, "window=%d", cfg->window);
  if (cfg->low_freq < 0 || cfg->low_freq >= cfg->fs / 2) return fail(MFCC_ERR_INVALID, "low_freq=%g", cfg->low_freq);
  if (cfg->high_freq != 0 && (cfg->high_freq <= cfg->low_freq || cfg->high_freq > cfg->fs / 2))
    return fail(MFCC_ERR_INVALID, "high_freq=%g outside (low_freq, fs/2]", cfg->high_freq);
  if (cfg->kernel != MFCC_KERNEL_AUTO && cfg->kernel != MFCC_KERNEL_GENERAL) return fail(MFCC_ERR_INVALID, "kernel=%d", cfg->kernel);
  if (cfg->frame_rule != 0 && cfg->frame_rule != 1) return fail(MFCC_ERR_INVALID, "frame_rule=%d", cfg->frame_rule);

  HIP_TRY(hipSetDevice(device));
  {
    // set-up time is where the static scratch of mfcc_cmvn_accumulate_static gets its address
    // on this device (that entry point resolves it itself if no plan was ever created here)
    double* scratch = nullptr;
    const int rc_s = cmvn_static_scratch(&scratch);
    if (rc_s) return rc_s;
  }
  mfcc_plan* p = new (std::nothrow) mfcc_plan();
  if (!p) return fail(MFCC_ERR_NOMEM, "mfcc_plan_create: out of host memory");
  p->cfg = *cfg;
  p->device = device;
  p->frame_size = size;
  p->frame_step = step;
  p->nfft = nfft;
  p->pre_hi = (float)cfg->preemph;
  p->pre_lo = (float)(cfg->preemph - (double)p->pre_hi);
  // spectrum rows in LDS hold 4|X|^2 (power) or 2|X| (magnitude); both scales are powers of two
  p->spec_scale = cfg->spectrum == MFCC_SPECTRUM_POWER ? 1.0f / (4.0f * nfft) : 0.5f;

  const bool g16k = size == mfcc::Geom16k::kSize && step == mfcc::Geom16k::kStep;
  const bool g8k = size == mfcc::Geom8k::kSize && step == mfcc::Geom8k::kStep;
  // 512-point FFT, or the 256-point FFT winfft derives for 200-sample frames (taken from the even bins)
  const bool g20 = size == mfcc::Geom20ms::kSize && step == mfcc::Geom20ms::kStep;  // frame_time = 20 at 16 kHz
  const bool fft_ok = nfft == mfcc::kNfft || (g8k && nfft == mfcc::kNfft / 2);
  p->plain_only = g20;
  bool specialised = cfg->kernel == MFCC_KERNEL_AUTO && (g16k || g8k || g20) && fft_ok && cfg->num_bins <= mfcc::kMaxMel &&
                     cfg->numcep <= mfcc::kMaxCep && cfg->pad_fac == 2 && !cfg->keep_c0;
  int rc = MFCC_OK;
  try {
    if (specialised) {
      rc = build_specialised(p);
      if (rc == 1) { specialised = false; rc = MFCC_OK; }
    }
    if (!specialised) p->plain_only = false;
    if (!specialised && !rc) rc = build_general(p);
    if (specialised && !rc) rc = allow_stacked_lds();
  } catch (const std::bad_alloc&) {
    rc = fail(MFCC_ERR_NOMEM, "mfcc_plan_create: out of host memory");
  }
  if (rc) {
    mfcc_plan_destroy(p);
    return rc;
  }
  *out = p;
  return MFCC_OK;
}

int mfcc_plan_destroy(mfcc_plan* p) {
  if (!p) return MFCC_OK;
  (void)hipFree(p->tw512);
  (void)hipFree(p->melW);
  (void)hipFree(p->dctW);
  (void)hipFree(p->window);
  (void)hipFree(p->pc);
  (void)hipFree(p->gc);
  (void)hipFree(p->g_tw);
  (void)hipFree(p->g_bank);
  (void)hipFree(p->g_band);
  (void)hipFree(p->g_dctl);
  (void)hipFree(p->g_window);
  delete p;
  return MFCC_OK;
}

int mfcc_plan_info(const mfcc_plan* p, int32_t* frame_size, int32_t* frame_step, int32_t* num_fft, int32_t* num_bins,
                   int32_t* numcep) {
  if (!p) return fail(MFCC_ERR_INVALID, "mfcc_plan_info: null plan");
  if (frame_size) *frame_size = p->frame_size;
  if (frame_step) *frame_step = p->frame_step;
  if (num_fft) *num_fft = p->nfft;
  if (num_bins) *num_bins = p->cfg.num_bins;
  if (numcep) *numcep = p->cfg.numcep;
  return MFCC_OK;
}

int32_t mfcc_plan_kernel(const mfcc_plan* p) {
  if (!p) return fail(MFCC_ERR_INVALID, "mfcc_plan_kernel: null plan");
  return p->general ? MFCC_KERNEL_GENERAL : MFCC_KERNEL_AUTO;
}

int32_t mfcc_plan_supports(const mfcc_plan* p, int32_t out_kind) {
  if (!p) return fail(MFCC_ERR_INVALID, "mfcc_plan_supports: null plan");
  if (out_kind < MFCC_OUT_MFCC39 || out_kind > MFCC_OUT_LOGSPEC) return fail(MFCC_ERR_INVALID, "out_kind=%d", out_kind);
  return (p->general && p->gen_no_delta && out_kind == MFCC_OUT_MFCC39) ? 0 : 1;
}

int32_t mfcc_out_width(const mfcc_plan* p, int32_t out_kind) {
  if (!p) return fail(MFCC_ERR_INVALID, "mfcc_out_width: null plan");
  switch (out_kind) {
    case MFCC_OUT_MFCC39: return 3 * p->cfg.numcep;
    case MFCC_OUT_STATIC: return p->cfg.numcep;
    case MFCC_OUT_LOGMEL: return p->cfg.num_bins + 1;
    case MFCC_OUT_SPEC:
    case MFCC_OUT_LOGSPEC: return p->nfft / 2 + 1;
  }
  return fail(MFCC_ERR_INVALID, "out_kind=%d", out_kind);
}

#ifdef MFCC_USE_SPLIT
// Tiles for the persistent workgroups of split_kernel.h.  A tile costs its batches (16 frames each, the halo
// frames at the sides that continue included); the tiles are dealt out in order, n_wg workgroups, each up to
// the same quota, and the tile that crosses a quota is cut at a batch boundary (its second part starts the next
// workgroup's list; a cut costs the 8 halo frames both parts compute).
static void split_schedule(const std::vector<mfcc::Tile>& tiles, const std::vector<int64_t>& frame_off, int n_wg,
                           std::vector<mfcc::Tile>* out_tiles, std::vector<int32_t>* out_off) {
  auto batches = [&](const mfcc::Tile& t) {
    const int64_t nf = frame_off[(size_t)t.utt + 1] - frame_off[(size_t)t.utt];
    const int64_t lo = std::max<int64_t>(0, (int64_t)t.t0 - mfcc::kHalo), hi = std::min<int64_t>(nf, (int64_t)t.t0 + t.nt + mfcc::kHalo);
    return (hi - lo + mfcc::kBatch - 1) / mfcc::kBatch;
  };
  int64_t rest = 0;  // batches of the tiles not yet dealt out (those behind the current one)
  for (const mfcc::Tile& t : tiles) rest += batches(t);
  out_tiles->clear();
  out_off->assign(1, 0);
  int64_t have = 0, quota = 0;
  // every workgroup's quota is its share of what is left, so that the batches the cuts add are spread too
  auto open = [&](int64_t left_now) {
    const int64_t wgs_left = n_wg - ((int64_t)out_off->size() - 1);
    quota = std::max<int64_t>(1, (left_now + wgs_left - 1) / std::max<int64_t>(1, wgs_left));
    have = 0;
  };
  auto close = [&](int64_t left_now) {
    out_off->push_back((int32_t)out_tiles->size());
    open(left_now);
  };
  open(rest);
  for (size_t i = 0; i < tiles.size(); ++i) {
    mfcc::Tile t = tiles[i];
    rest -= batches(t);
    for (;;) {
      const bool last_wg = (int)out_off->size() == n_wg;  // the last workgroup takes what is left
      const int64_t nbt = batches(t);
      if (last_wg || have + nbt <= quota) {
        out_tiles->push_back(t);
        have += nbt;
        if (!last_wg && have >= quota) close(rest);
        break;
      }
      const int64_t room = quota - have;
      const int64_t left = t.t0 > 0 ? mfcc::kHalo : 0;
      const int64_t nt1 = room * mfcc::kBatch - left - mfcc::kHalo;  // its frames + both halos = room whole batches
      if (room >= 2 && nt1 >= 1 && t.nt - nt1 >= mfcc::kBatch) {
        out_tiles->push_back(mfcc::Tile{t.utt, t.t0, (int32_t)nt1, 0});
        t = mfcc::Tile{t.utt, (int32_t)(t.t0 + nt1), (int32_t)(t.nt - nt1), 0};
        close(rest + batches(t));
      } else if (have > 0) {
        close(rest + nbt);
      } else {  // (cannot happen: an empty workgroup takes any tile; kept so that the loop always ends)
        out_tiles->push_back(t);
        close(rest);
        break;
      }
    }
  }
  while ((int)out_off->size() < n_wg + 1) out_off->push_back((int32_t)out_tiles->size());
}
#endif

static int layout_create_impl(const mfcc_plan* plan, const int64_t* lengths_host, int64_t n_utts, mfcc_layout** out) {
  if (!plan || !out || n_utts < 0 || (n_utts > 0 && !lengths_host)) return fail(MFCC_ERR_INVALID, "mfcc_layout_create: bad argument");
  if (n_utts > INT32_MAX) return fail(MFCC_ERR_UNSUPPORTED, "more than 2^31-1 utterances in one layout");
  *out = nullptr;
  mfcc_layout* l = new (std::nothrow) mfcc_layout();
  if (!l) return fail(MFCC_ERR_NOMEM, "mfcc_layout_create: out of host memory");
  l->device = plan->device;
  l->n_utts = n_utts;
  l->frame_size = plan->frame_size;
  l->frame_step = plan->frame_step;
  l->frame_rule = plan->cfg.frame_rule;
  l->general_cut = plan->general;
  l->sample_off.assign(n_utts + 1, 0);
  l->frame_off.assign(n_utts + 1, 0);
  std::vector<mfcc::Tile> tiles;
  const int64_t tile_max = mfcc::tile_frames_for(plan->cfg.numcep);
  for (int64_t u = 0; u < n_utts; ++u) {
    const int64_t len = lengths_host[u];
    if (len < 1) {
      delete l;
      return fail(MFCC_ERR_INVALID, "utterance %lld has length %lld (the reference needs >= 1 sample: signal[0], mfcc-self5.py:68)",
                  (long long)u, (long long)len);
    }
    const int64_t nf = plan->cfg.frame_rule == 1
                           ? std::max<int64_t>(0, (len - (plan->frame_size - plan->frame_step)) / plan->frame_step)
                           : mfcc::frames_for_length(len, plan->frame_size, plan->frame_step);
    if (nf > INT32_MAX) {
      delete l;
      return fail(MFCC_ERR_UNSUPPORTED, "utterance %lld has more than 2^31-1 frames", (long long)u);
    }
    l->sample_off[u + 1] = l->sample_off[u] + len;
    l->frame_off[u + 1] = l->frame_off[u] + nf;
    // greedy tiling: every tile computes a whole number of 16-frame batches where it can
    // (tile frames + 4 halo frames on each side that continues)
    for (int64_t t = 0; t < nf;) {
      int64_t nt = nf - t;
      if (plan->general) {
        nt = std::min<int64_t>(nt, plan->gen_nt_max);
      } else {
        const int64_t left = t > 0 ? mfcc::kHalo : 0;
        if (nt + left > tile_max) nt = tile_max - left - mfcc::kHalo;
      }
      tiles.push_back(mfcc::Tile{(int32_t)u, (int32_t)t, (int32_t)nt, 0});
      l->max_nt = std::max<int32_t>(l->max_nt, (int32_t)nt);
      t += nt;
    }
  }
  l->total_samples = l->sample_off[n_utts];
  l->total_frames = l->frame_off[n_utts];
  hipError_t e = hipSetDevice(plan->device);
  int rc = e == hipSuccess ? MFCC_OK : fail(MFCC_ERR_HIP, "hipSetDevice: %s", hipGetErrorString(e));
  if (!rc && !plan->general) {
    // Last round of the grid.  The fused kernel runs three workgroups per CU at a time; when the tile
    // count leaves a few tiles over whole rounds, those would run on a few CUs while the rest of
    // the GPU has nothing left (2 % of the launch for 10 000 equal tiles on 768 places).  They are
    // the last in dispatch order: cut each into pieces of about five batches, so that the
    // remainder spreads over four times as many CUs.
    int n_cu = 0;
    if (hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, plan->device) != hipSuccess) n_cu = 0;
    const int64_t places = (int64_t)3 * n_cu, n = (int64_t)tiles.size(), rest = places > 0 ? n % places : 0;
    if (n > places && rest > 0 && rest <= places / 8) {
      std::vector<mfcc::Tile> cut;
      for (int64_t i = n - rest; i < n; ++i) {
        const mfcc::Tile big = tiles[(size_t)i];
        const int64_t end = (int64_t)big.t0 + big.nt;
        for (int64_t t = big.t0; t < end;) {
          // five whole batches including the halo frames on the sides that continue
          const int64_t left = t > 0 ? mfcc::kHalo : 0;
          int64_t nt = 5 * mfcc::kBatch - left - mfcc::kHalo;
          if (t + nt + mfcc::kBatch >= end) nt = end - t;  // no sliver at the end
          cut.push_back(mfcc::Tile{big.utt, (int32_t)t, (int32_t)nt, 0});
          t += nt;
        }
      }
      tiles.resize((size_t)(n - rest));
      tiles.insert(tiles.end(), cut.begin(), cut.end());
    }
  }
  l->n_tiles = (int64_t)tiles.size();
  if (l->n_tiles > INT32_MAX) {
    mfcc_layout_destroy(l);
    return fail(MFCC_ERR_UNSUPPORTED, "layout needs more than 2^31-1 workgroups; split the batch");
  }
#ifdef MFCC_USE_SPLIT
  if (!rc && !plan->general && !tiles.empty()) {
    int n_cu = 0;
    if (hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, plan->device) != hipSuccess || n_cu < 1) n_cu = 256;
    // two eight-wave workgroups fit a CU; four lists per CU (two rounds) measured best: 1.075 ms against 1.10 ms
    // with two and 1.07 ms with one tile per list (MFCC_DEBUG_WG_PER_CU sets the number for experiments)
    const char* const dbg_wg = getenv("MFCC_DEBUG_WG_PER_CU");
    const int per_cu = dbg_wg && atoi(dbg_wg) > 0 ? atoi(dbg_wg) : 4;
    l->n_wg = (int32_t)std::min<int64_t>(per_cu * (int64_t)n_cu, l->n_tiles);
    std::vector<mfcc::Tile> wg_tiles;
    std::vector<int32_t> wg_off;
    split_schedule(tiles, l->frame_off, l->n_wg, &wg_tiles, &wg_off);
    rc = upload(wg_tiles, &l->d_wg_tiles);
    if (!rc) rc = upload(wg_off, &l->d_wg_off);
  }
#endif
  if (!rc) rc = upload(l->sample_off, &l->d_sample_off);
  if (!rc) rc = upload(l->frame_off, &l->d_frame_off);
  if (!rc) rc = upload(tiles, &l->d_tiles);
  if (rc) {
    mfcc_layout_destroy(l);
    return rc;
  }
  *out = l;
  return MFCC_OK;
}

int mfcc_layout_create(const mfcc_plan* plan, const int64_t* lengths_host, int64_t n_utts, mfcc_layout** out) {
  // the host-side tables are std::vectors: no C++ exception may cross the C boundary
  try {
    return layout_create_impl(plan, lengths_host, n_utts, out);
  } catch (const std::bad_alloc&) {
    return fail(MFCC_ERR_NOMEM, "mfcc_layout_create: out of host memory");
  }
}

int mfcc_layout_create_frames(int device, const int64_t* frame_counts_host, int64_t n_utts, mfcc_layout** out) {
  if (!out || n_utts < 0 || (n_utts > 0 && !frame_counts_host)) return fail(MFCC_ERR_INVALID, "mfcc_layout_create_frames: bad argument");
  if (n_utts > INT32_MAX) return fail(MFCC_ERR_UNSUPPORTED, "more than 2^31-1 utterances in one layout");
  *out = nullptr;
  try {
    mfcc_layout* l = new (std::nothrow) mfcc_layout();
    if (!l) return fail(MFCC_ERR_NOMEM, "mfcc_layout_create_frames: out of host memory");
    l->device = device;
    l->n_utts = n_utts;
    l->sample_off.assign(n_utts + 1, 0);
    l->frame_off.assign(n_utts + 1, 0);
    for (int64_t u = 0; u < n_utts; ++u) {
      if (frame_counts_host[u] < 0) {
        delete l;
        return fail(MFCC_ERR_INVALID, "utterance %lld has a negative frame count", (long long)u);
      }
      l->frame_off[u + 1] = l->frame_off[u] + frame_counts_host[u];
    }
    l->total_samples = 0;
    l->total_frames = l->frame_off[n_utts];
    l->n_tiles = 0;
    l->frames_only = true;
    hipError_t e = hipSetDevice(device);
    int rc = e == hipSuccess ? MFCC_OK : fail(MFCC_ERR_HIP, "hipSetDevice: %s", hipGetErrorString(e));
    if (!rc) rc = upload(l->frame_off, &l->d_frame_off);
    if (rc) {
      mfcc_layout_destroy(l);
      return rc;
    }
    *out = l;
    return MFCC_OK;
  } catch (const std::bad_alloc&) {
    return fail(MFCC_ERR_NOMEM, "mfcc_layout_create_frames: out of host memory");
  }
}

int mfcc_layout_destroy(mfcc_layout* l) {
  if (!l) return MFCC_OK;
  (void)hipFree(l->d_sample_off);
  (void)hipFree(l->d_frame_off);
  (void)hipFree(l->d_tiles);
  (void)hipFree(l->d_wg_tiles);
  (void)hipFree(l->d_wg_off);
  delete l;
  return MFCC_OK;
}

int mfcc_layout_info(const mfcc_layout* l, int64_t* n_utts, int64_t* total_samples, int64_t* total_frames) {
  if (!l) return fail(MFCC_ERR_INVALID, "mfcc_layout_info: null layout");
  if (n_utts) *n_utts = l->n_utts;
  if (total_samples) *total_samples = l->total_samples;
  if (total_frames) *total_frames = l->total_frames;
  return MFCC_OK;
}

int mfcc_layout_offsets_host(const mfcc_layout* l, int64_t* sample_offsets_host, int64_t* frame_offsets_host) {
  if (!l) return fail(MFCC_ERR_INVALID, "mfcc_layout_offsets_host: null layout");
  if (sample_offsets_host) memcpy(sample_offsets_host, l->sample_off.data(), l->sample_off.size() * sizeof(int64_t));
  if (frame_offsets_host) memcpy(frame_offsets_host, l->frame_off.data(), l->frame_off.size() * sizeof(int64_t));
  return MFCC_OK;
}

// mfcc_forward; tile_sums != nullptr: the MFCC-39 kind that also leaves the column sums of every tile there
static int forward_rows(const mfcc_plan* plan, const mfcc_layout* layout, const void* audio, int32_t audio_dtype, float* out,
                        int64_t out_ld, int32_t out_kind, float* tile_sums, void* stream) {
  if (!plan || !layout) return fail(MFCC_ERR_INVALID, "mfcc_forward: null plan or layout");
  if (plan->device != layout->device) return fail(MFCC_ERR_INVALID, "plan and layout live on different devices");
  if (layout->frames_only) return fail(MFCC_ERR_INVALID, "mfcc_forward: the layout holds frame counts only (mfcc_layout_create_frames)");
  {
    const int rc_l = layout_fits(plan, layout);
    if (rc_l) return rc_l;
  }
  if (audio_dtype != MFCC_AUDIO_I16 && audio_dtype != MFCC_AUDIO_F32) return fail(MFCC_ERR_INVALID, "audio_dtype=%d", audio_dtype);
  const int32_t width = mfcc_out_width(plan, out_kind);
  if (width < 0) return width;
  if (out_ld < width) return fail(MFCC_ERR_INVALID, "out_ld=%lld < row width %d", (long long)out_ld, width);
  // the specialised kernel addresses the rows of a tile (up to 19 batches + halo) with 32-bit offsets from the tile's first row
  if (!plan->general && out_ld * (int64_t)(mfcc::tile_frames_for(plan->cfg.numcep) + 2 * mfcc::kBatch) >= ((int64_t)1 << 31))
    return fail(MFCC_ERR_UNSUPPORTED, "out_ld=%lld: rows more than %lld floats apart are beyond the specialised kernel's 32-bit row offsets",
                (long long)out_ld, (long long)((((int64_t)1 << 31) - 1) / (mfcc::tile_frames_for(plan->cfg.numcep) + 2 * mfcc::kBatch)));
  if (plan->general && plan->gen_no_delta && out_kind == MFCC_OUT_MFCC39)
    return fail(MFCC_ERR_UNSUPPORTED, "num_fft=%d, num_bins=%d, numcep=%d, pad_fac=%d: the delta rows of a tile do not fit a workgroup's LDS; "
                "take MFCC_OUT_STATIC and apply mfcc_delta per utterance", plan->nfft, plan->cfg.num_bins, plan->cfg.numcep, plan->cfg.pad_fac);
  if (layout->n_utts == 0 || layout->n_tiles == 0) return MFCC_OK;
  if (!out || (!audio && layout->total_samples > 0)) return fail(MFCC_ERR_INVALID, "mfcc_forward: null device pointer");

  mfcc::FusedParams fp;
  fp.audio = audio;
  fp.sample_off = layout->d_sample_off;
  fp.sample_len = nullptr;
  fp.frame_off = layout->d_frame_off;
  fp.tiles = layout->d_tiles;
  fp.out = out;
  fp.out_ld = out_ld;
  fp.total_elems = layout->total_samples;
  if (layout->total_samples == 0) {
    // only empty utterances (one all-zero frame each): the kernel's clamped staging loads still
    // need 16 readable bytes, which the layout's own offset array provides
    fp.audio = layout->d_sample_off;
    fp.total_elems = 1;
  }
  fp.pc = plan->pc;
  fp.stack_n = 0;
  fp.tile_sums = tile_sums;
  fp.wg_tiles = layout->d_wg_tiles;
  fp.wg_off = layout->d_wg_off;
#ifdef MFCC_DIAG
  fp.dbg = g_dbg;
#endif

  launch_fused(plan, fp, layout->n_tiles, tile_sums ? kLaunchMfcc39Sums : out_kind, audio_dtype == MFCC_AUDIO_I16, (hipStream_t)stream,
               layout->n_wg);
  HIP_TRY(hipGetLastError());
  return MFCC_OK;
}

int mfcc_forward(const mfcc_plan* plan, const mfcc_layout* layout, const void* audio, int32_t audio_dtype, float* out,
                 int64_t out_ld, int32_t out_kind, void* stream) {
  return forward_rows(plan, layout, audio, audio_dtype, out, out_ld, out_kind, nullptr, stream);
}

int64_t mfcc_forward_sums_workspace(const mfcc_plan* plan, const mfcc_layout* layout) {
  if (!plan || !layout) return fail(MFCC_ERR_INVALID, "mfcc_forward_sums_workspace: null plan or layout");
  if (plan->general) return fail(MFCC_ERR_UNSUPPORTED, "mfcc_forward_sums: plans of the specialised kernel only; use mfcc_forward + mfcc_cmvn_accumulate");
  return (int64_t)(layout->n_tiles > 0 ? layout->n_tiles : 1) * mfcc::kTileSums * (int64_t)sizeof(float);
}

int mfcc_forward_sums(const mfcc_plan* plan, const mfcc_layout* layout, const void* audio, int32_t audio_dtype, float* out,
                      int64_t out_ld, double* sums, void* workspace, void* stream) {
  if (!plan || !layout) return fail(MFCC_ERR_INVALID, "mfcc_forward_sums: null plan or layout");
  if (plan->general) return fail(MFCC_ERR_UNSUPPORTED, "mfcc_forward_sums: plans of the specialised kernel only; use mfcc_forward + mfcc_cmvn_accumulate");
  if (!sums || !workspace) return fail(MFCC_ERR_INVALID, "mfcc_forward_sums: null sums or workspace");
  if ((uintptr_t)workspace & 3) return fail(MFCC_ERR_INVALID, "mfcc_forward_sums: the workspace must be 4-byte aligned");
  const int rc = forward_rows(plan, layout, audio, audio_dtype, out, out_ld, MFCC_OUT_MFCC39, (float*)workspace, stream);
  if (rc) return rc;
  if (layout->n_utts == 0 || layout->n_tiles == 0) return MFCC_OK;
  hipLaunchKernelGGL(mfcc::tile_sums_finish_kernel, dim3(6 * plan->cfg.numcep), dim3(256), 0, (hipStream_t)stream, (const float*)workspace,
                     (const mfcc::Tile*)layout->d_tiles, layout->n_tiles, (int)plan->cfg.numcep, layout->total_frames, sums);
  HIP_TRY(hipGetLastError());
  return MFCC_OK;
}

int mfcc_forward_stacked(const mfcc_plan* plan, const mfcc_layout* layout, const void* audio, int32_t audio_dtype, int32_t framenum,
                         float* out, int32_t out_kind, void* stream) {
  if (!plan || !layout) return fail(MFCC_ERR_INVALID, "mfcc_forward_stacked: null plan or layout");
  if (plan->device != layout->device) return fail(MFCC_ERR_INVALID, "plan and layout live on different devices");
  if (layout->frames_only) return fail(MFCC_ERR_INVALID, "mfcc_forward_stacked: the layout holds frame counts only (mfcc_layout_create_frames)");
  if (plan->general || plan->plain_only || (plan->nfft != mfcc::kNfft && out_kind != MFCC_OUT_MFCC39))
    return fail(MFCC_ERR_UNSUPPORTED, "mfcc_forward_stacked: built for 512-point plans of the specialised kernel with 400/160- or "
                "200/80-sample frames (this one: framing %d/%d, %d-point FFT%s); use mfcc_forward + mfcc_stack_context",
                plan->frame_size, plan->frame_step, plan->nfft, plan->general ? ", general kernel" : "");
  {
    const int rc_l = layout_fits(plan, layout);
    if (rc_l) return rc_l;
  }
  if (audio_dtype != MFCC_AUDIO_I16 && audio_dtype != MFCC_AUDIO_F32) return fail(MFCC_ERR_INVALID, "audio_dtype=%d", audio_dtype);
  if (out_kind != MFCC_OUT_SPEC && out_kind != MFCC_OUT_LOGSPEC && out_kind != MFCC_OUT_MFCC39)
    return fail(MFCC_ERR_INVALID, "mfcc_forward_stacked: out_kind must be MFCC_OUT_SPEC, MFCC_OUT_LOGSPEC or MFCC_OUT_MFCC39");
  if (out_kind == MFCC_OUT_MFCC39 && framenum < 3)
    return fail(MFCC_ERR_INVALID, "mfcc_forward_stacked: MFCC_OUT_MFCC39 takes framenum >= 3 (framenum 1 is mfcc_forward)");
  if (framenum < 1 || framenum % 2 == 0)
    return fail(MFCC_ERR_INVALID, "framenum=%d: use an odd number of frames (MLP_5hidlayer_40.py:261-262 warns the same)", framenum);
  if (framenum > mfcc::kStackMaxN)
    return fail(MFCC_ERR_UNSUPPORTED, "framenum=%d: the one-launch path takes odd framenum up to %d (the window has to fit the ring of "
                "spectrum rows in LDS); use mfcc_forward + mfcc_stack_context", framenum, mfcc::kStackMaxN);
  if (layout->n_utts == 0 || layout->n_tiles == 0) return MFCC_OK;
  if (!out || !audio) return fail(MFCC_ERR_INVALID, "mfcc_forward_stacked: null device pointer");
  if ((uintptr_t)out & 15) return fail(MFCC_ERR_INVALID, "mfcc_forward_stacked: out must be 16-byte aligned");

  mfcc::FusedParams fp;
  fp.audio = audio;
  fp.sample_off = layout->d_sample_off;
  fp.sample_len = nullptr;
  fp.frame_off = layout->d_frame_off;
  fp.tiles = layout->d_tiles;
  fp.out = out;
  fp.out_ld = out_kind == MFCC_OUT_MFCC39 ? (int64_t)framenum * 3 * plan->cfg.numcep : (int64_t)framenum * (plan->nfft / 2 + 1);
  fp.total_elems = layout->total_samples;
  fp.pc = plan->pc;
  fp.stack_n = framenum;
  fp.tile_sums = nullptr;
  fp.wg_tiles = nullptr;
  fp.wg_off = nullptr;
#ifdef MFCC_DIAG
  fp.dbg = nullptr;
#endif
  launch_fused(plan, fp, layout->n_tiles,
               out_kind == MFCC_OUT_MFCC39 ? kLaunchMfcc39Stack : (out_kind == MFCC_OUT_SPEC ? kLaunchSpecStack : kLaunchLogspecStack),
               audio_dtype == MFCC_AUDIO_I16, (hipStream_t)stream);
  HIP_TRY(hipGetLastError());
  return MFCC_OK;
}

int mfcc_preemphasis(const void* signal, int32_t dtype, int64_t n, double coeff, float* out, void* stream) {
  if (n < 0 || (n > 0 && (!signal || !out))) return fail(MFCC_ERR_INVALID, "mfcc_preemphasis: bad argument");
  if (n == 0) return MFCC_OK;
  const float hi = (float)coeff, lo = (float)(coeff - (double)hi);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MFCC_AUDIO_I16)
    hipLaunchKernelGGL(mfcc::preemph_kernel<int16_t>, dim3(blocks_for(n, 256)), dim3(256), 0, st, (const int16_t*)signal, n, hi, lo, out);
  else if (dtype == MFCC_AUDIO_F32)
    hipLaunchKernelGGL(mfcc::preemph_kernel<float>, dim3(blocks_for(n, 256)), dim3(256), 0, st, (const float*)signal, n, hi, lo, out);
  else
    return fail(MFCC_ERR_INVALID, "dtype=%d", dtype);
  HIP_TRY(hipGetLastError());
  return MFCC_OK;
}

int mfcc_melspec(const float* spec, int64_t n_frames, int32_t n_bins, const float* melbank, int32_t nmel, int32_t floor_mel,
                 float* logmel, float* loge, void* stream) {
  if (n_frames < 0 || n_bins < 1 || nmel < 1) return fail(MFCC_ERR_INVALID, "mfcc_melspec: bad shape");
  if (n_frames == 0) return MFCC_OK;
  if (!spec || !melbank || !logmel || !loge) return fail(MFCC_ERR_INVALID, "mfcc_melspec: null pointer");
  if (n_frames > INT32_MAX) return fail(MFCC_ERR_UNSUPPORTED, "mfcc_melspec: more than 2^31-1 frames per call");
  hipLaunchKernelGGL(mfcc::melspec_kernel, dim3((unsigned)n_frames), dim3(64), 0, (hipStream_t)stream, spec, n_bins, melbank, nmel,
                     floor_mel, logmel, loge);
  HIP_TRY(hipGetLastError());
  return MFCC_OK;
}

int mfcc_dct(const float* logmel, int64_t n_frames, int32_t nmel, int32_t numcep, float* out, void* stream) {
  if (n_frames < 0 || nmel < 1 || numcep < 1 || numcep > nmel) return fail(MFCC_ERR_INVALID, "mfcc_dct: bad shape");
  if (n_frames == 0) return MFCC_OK;
  if (!logmel || !out) return fail(MFCC_ERR_INVALID, "mfcc_dct: null pointer");
  const int64_t total = n_frames * numcep;
  hipLaunchKernelGGL(mfcc::dct_kernel, dim3(blocks_for(total, 256)), dim3(256), 0, (hipStream_t)stream, logmel, n_frames, nmel, numcep, out);
  HIP_TRY(hipGetLastError());
  return MFCC_OK;
}

int mfcc_lifter(const float* cep, int64_t n_frames, int32_t ncoeff, double lift_factor, float* out, void* stream) {
  if (n_frames < 0 || ncoeff < 1) return fail(MFCC_ERR_INVALID, "mfcc_lifter: bad shape");
  if (n_frames == 0) return MFCC_OK;
  if (!cep || !out) return fail(MFCC_ERR_INVALID, "mfcc_lifter: null pointer");
  const int64_t total = n_frames * ncoeff;
  hipLaunchKernelGGL(mfcc::lifter_kernel, dim3(blocks_for(total, 256)), dim3(256), 0, (hipStream_t)stream, cep, n_frames, ncoeff,
                     lift_factor, out);
  HIP_TRY(hipGetLastError());
  return MFCC_OK;
}

int mfcc_delta(const float* feats, int64_t n_frames, int32_t ncol, int32_t pad_fac, float* out, void* stream) {
  if (n_frames < 0 || ncol < 1 || pad_fac < 1) return fail(MFCC_ERR_INVALID, "mfcc_delta: bad shape");
  if (n_frames == 0) return MFCC_OK;
  if (!feats || !out) return fail(MFCC_ERR_INVALID, "mfcc_delta: null pointer");
  int denom = 0;
  for (int i = 1; i <= pad_fac; ++i) denom += 2 * i * i;  // mfcc-self5.py:172
  const int64_t total = n_frames * ncol;
  hipLaunchKernelGGL(mfcc::delta_kernel, dim3(blocks_for(total, 256)), dim3(256), 0, (hipStream_t)stream, feats, n_frames, ncol, pad_fac,
                     1.0f / (float)denom, out);
  HIP_TRY(hipGetLastError());
  return MFCC_OK;
}

int mfcc_copy_cols(const float* src, int64_t src_ld, float* dst, int64_t dst_ld, int64_t n_rows, int32_t ncol, void* stream) {
  if (n_rows < 0 || ncol < 1 || src_ld < ncol || dst_ld < ncol) return fail(MFCC_ERR_INVALID, "mfcc_copy_cols: bad shape");
  if (n_rows == 0) return MFCC_OK;
  if (!src || !dst) return fail(MFCC_ERR_INVALID, "mfcc_copy_cols: null pointer");
  const int64_t total = n_rows * ncol;
  hipLaunchKernelGGL(mfcc::copy_cols_kernel, dim3(blocks_for(total, 256)), dim3(256), 0, (hipStream_t)stream, src, src_ld, dst, dst_ld,
                     n_rows, ncol);
  HIP_TRY(hipGetLastError());
  return MFCC_OK;
}

int mfcc_fir(const mfcc_layout* layout, const void* audio, int32_t audio_dtype, const float* taps_host, int32_t ntaps, int32_t delay,
             float* out, void* stream) {
  if (!layout || !taps_host) return fail(MFCC_ERR_INVALID, "mfcc_fir: null argument");
  if (layout->frames_only) return fail(MFCC_ERR_INVALID, "mfcc_fir: the layout holds frame counts only (mfcc_layout_create_frames)");
  if (ntaps < 1 || ntaps > 32 || delay < 0 || delay >= ntaps) return fail(MFCC_ERR_INVALID, "mfcc_fir: 1..32 taps, 0 <= delay < ntaps");
  if (audio_dtype != MFCC_AUDIO_I16 && audio_dtype != MFCC_AUDIO_F32) return fail(MFCC_ERR_INVALID, "audio_dtype=%d", audio_dtype);
  if (layout->total_samples == 0) return MFCC_OK;
  if (!audio || !out) return fail(MFCC_ERR_INVALID, "mfcc_fir: null device pointer");
  mfcc::FirTaps t;
  memset(&t, 0, sizeof(t));
  memcpy(t.w, taps_host, ntaps * sizeof(float));
  t.n = ntaps;
  t.delay = delay;
  const int64_t nblk = (layout->total_samples + 1023) / 1024;
  if (nblk > INT32_MAX) return fail(MFCC_ERR_UNSUPPORTED, "mfcc_fir: too many samples per call");
  if (audio_dtype == MFCC_AUDIO_I16)
    hipLaunchKernelGGL(mfcc::fir_kernel<int16_t>, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, (const int16_t*)audio,
                       layout->d_sample_off, (int)layout->n_utts, layout->total_samples, t, out);
  else
    hipLaunchKernelGGL(mfcc::fir_kernel<float>, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, (const float*)audio,
                       layout->d_sample_off, (int)layout->n_utts, layout->total_samples, t, out);
  HIP_TRY(hipGetLastError());
  return MFCC_OK;
}

int mfcc_stack_context(const mfcc_layout* layout, const float* feats, int64_t feats_ld, int32_t ncol, int32_t framenum,
                       int32_t out_layout, const int64_t* row_index, int64_t n_out, float* out, void* stream) {
  if (!layout) return fail(MFCC_ERR_INVALID, "mfcc_stack_context: null layout");
  if (ncol < 1 || feats_ld < ncol || framenum < 1) return fail(MFCC_ERR_INVALID, "mfcc_stack_context: bad shape");
  if (framenum % 2 == 0)
    return fail(MFCC_ERR_INVALID, "framenum=%d: use an odd number of frames (MLP_5hidlayer_40.py:261-262 warns the same)", framenum);
  if (out_layout != MFCC_STACK_FRAME_MAJOR && out_layout != MFCC_STACK_TILE) return fail(MFCC_ERR_INVALID, "out_layout=%d", out_layout);
  if (!row_index) n_out = layout->total_frames;
  if (n_out < 0) return fail(MFCC_ERR_INVALID, "mfcc_stack_context: n_out=%lld", (long long)n_out);
  if (layout->total_frames == 0 || n_out == 0) return MFCC_OK;
  if (!feats || !out) return fail(MFCC_ERR_INVALID, "mfcc_stack_context: null pointer");
  if (layout->total_frames > INT32_MAX || n_out > INT32_MAX) return fail(MFCC_ERR_UNSUPPORTED, "mfcc_stack_context: more than 2^31-1 rows per call");
  hipStream_t st = (hipStream_t)stream;
  if (out_layout == MFCC_STACK_TILE || row_index) {
    const size_t smem = (size_t)framenum * (ncol | 1) * sizeof(float);
    if (smem > 64 * 1024) return fail(MFCC_ERR_UNSUPPORTED, "mfcc_stack_context: framenum * ncol = %d exceeds the 16384 values a window may hold "
                                      "in this layout / order", framenum * ncol);
    hipLaunchKernelGGL(mfcc::stack_context_gather_kernel, dim3((unsigned)n_out), dim3(256), smem, st, feats, feats_ld, ncol, framenum,
                       layout->d_frame_off, (int)layout->n_utts, layout->total_frames, row_index, out_layout == MFCC_STACK_TILE ? 1 : 0, out);
  } else if (((uintptr_t)out & 15) == 0 && (int64_t)framenum * ncol * mfcc::kStackRows < (1 << 24)) {
    const int64_t nblk = ((layout->total_frames + mfcc::kStackRows - 1) / mfcc::kStackRows + 7) / 8 * 8;  // whole rounds over the 8 XCDs
    hipLaunchKernelGGL(mfcc::stack_context_rows_kernel, dim3((unsigned)nblk), dim3(256), 0, st, feats, feats_ld, ncol,
                       framenum, layout->d_frame_off, (int)layout->n_utts, out, layout->total_frames);
  } else {
    hipLaunchKernelGGL(mfcc::stack_context_kernel, dim3((unsigned)layout->total_frames), dim3(256), 0, st, feats, feats_ld,
                       ncol, framenum, layout->d_frame_off, (int)layout->n_utts, out);
  }
  HIP_TRY(hipGetLastError());
  return MFCC_OK;
}

int mfcc_cmvn_accumulate(const float* feats, int64_t feats_ld, int64_t n_rows, int32_t ncol, double* sums, void* stream) {
  if (n_rows < 0 || ncol < 1 || ncol > kCmvnMaxCols || feats_ld < ncol) return fail(MFCC_ERR_INVALID, "mfcc_cmvn_accumulate: bad shape");
  if (n_rows == 0) return MFCC_OK;
  if (!feats || !sums) return fail(MFCC_ERR_INVALID, "mfcc_cmvn_accumulate: null pointer");
  int nblk;
  int64_t rpb;
  cmvn_split(n_rows, &nblk, &rpb);
  double* partial = nullptr;
  const int rc = cmvn_workspace((hipStream_t)stream, &partial);
  if (rc) return rc;
  return launch_cmvn(feats, feats_ld, n_rows, ncol, sums, partial, nblk, rpb, (hipStream_t)stream);
}

int mfcc_cmvn_release(void) {
  int device = 0;
  HIP_TRY(hipGetDevice(&device));
  std::lock_guard<std::mutex> lock(g_ws_mutex);
  for (size_t i = 0; i < g_ws.size();)
    if (g_ws[i].device == device) {
      (void)hipFree(g_ws[i].ptr);
      g_ws.erase(g_ws.begin() + (ptrdiff_t)i);
    } else {
      ++i;
    }
  return MFCC_OK;
}

int mfcc_cmvn_accumulate_static(const float* feats, int64_t feats_ld, int64_t n_rows, int32_t ncol, double* sums, void* stream) {
  if (n_rows < 0 || ncol < 1 || feats_ld < ncol) return fail(MFCC_ERR_INVALID, "mfcc_cmvn_accumulate_static: bad shape");
  if (ncol > kCmvnStaticCols)
    return fail(MFCC_ERR_UNSUPPORTED, "mfcc_cmvn_accumulate_static: %d columns, the static scratch holds at most %d "
                "(use mfcc_cmvn_accumulate)", ncol, kCmvnStaticCols);
  if (n_rows == 0) return MFCC_OK;
  if (!feats || !sums) return fail(MFCC_ERR_INVALID, "mfcc_cmvn_accumulate_static: null pointer");
  int nblk;
  int64_t rpb;
  cmvn_split(n_rows, &nblk, &rpb);
  double* partial = nullptr;
  hipEvent_t turn = nullptr;
  const int rc = cmvn_static_scratch(&partial, &turn);
  if (rc) return rc;
  // Calls from different streams share the one array: outside stream capture each call waits for
  // the call before it on this device and leaves its own mark (two stream operations, no
  // allocation, no host wait).  Under capture neither is issued -- an event recorded into a graph
  // cannot order eager work -- and the captured call is ordered by its graph alone.
  const hipStream_t st = (hipStream_t)stream;
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  HIP_TRY(hipStreamIsCapturing(st, &cap));
  const bool eager = cap == hipStreamCaptureStatusNone;
  if (eager) HIP_TRY(hipStreamWaitEvent(st, turn, 0));
  // every partial the second stage reads is written by the first stage of this very call, so
  // nothing of an earlier call's contents has to be cleared
  const int rc_l = launch_cmvn(feats, feats_ld, n_rows, ncol, sums, partial, nblk, rpb, st);
  if (rc_l) return rc_l;
  if (eager) HIP_TRY(hipEventRecord(turn, st));
  return MFCC_OK;
}

int mfcc_cmvn_apply(float* feats, int64_t feats_ld, int64_t n_rows, int32_t ncol, const float* mm, const float* vv, void* stream) {
  if (n_rows < 0 || ncol < 1 || feats_ld < ncol) return fail(MFCC_ERR_INVALID, "mfcc_cmvn_apply: bad shape");
  if (n_rows == 0) return MFCC_OK;
  if (!feats || !mm || !vv) return fail(MFCC_ERR_INVALID, "mfcc_cmvn_apply: null pointer");
  if (ncol > 4096) return fail(MFCC_ERR_UNSUPPORTED, "mfcc_cmvn_apply: more than 4096 columns");
  {
    // rows back to back: 16-byte pieces, about 8 per thread
    const int64_t n_elems = n_rows * ncol;
    const int flat = cmvn_flat_blocks(feats, feats_ld, ncol, (n_elems / 4 + 256 * 8 - 1) / (256 * 8), 16384);
    if (flat) {
      hipLaunchKernelGGL(mfcc::cmvn_apply_flat_kernel, dim3((unsigned)flat), dim3(256), 0, (hipStream_t)stream, feats, n_elems, ncol, mm, vv);
      HIP_TRY(hipGetLastError());
      return MFCC_OK;
    }
  }
  const int rows_per_block = std::max(1, 16384 / ncol);
  const int64_t nblk = (n_rows + rows_per_block - 1) / rows_per_block;
  if (nblk > INT32_MAX) return fail(MFCC_ERR_UNSUPPORTED, "mfcc_cmvn_apply: too many rows per call");
  hipLaunchKernelGGL(mfcc::cmvn_apply_kernel, dim3((unsigned)nblk), dim3(256), 2 * ncol * sizeof(float), (hipStream_t)stream, feats,
                     feats_ld, n_rows, ncol, mm, vv, rows_per_block);
  HIP_TRY(hipGetLastError());
  return MFCC_OK;
}

int mfcc_utt_scalar_norm(const mfcc_layout* layout, float* feats, int64_t feats_ld, int32_t ncol, void* stream) {
  if (!layout) return fail(MFCC_ERR_INVALID, "mfcc_utt_scalar_norm: null layout");
  if (ncol < 1 || ncol > 256 || feats_ld < ncol) return fail(MFCC_ERR_INVALID, "mfcc_utt_scalar_norm: bad shape (1 <= ncol <= 256)");
  if (layout->n_utts == 0) return MFCC_OK;
  if (!feats) return fail(MFCC_ERR_INVALID, "mfcc_utt_scalar_norm: null pointer");
  hipLaunchKernelGGL(mfcc::utt_scalar_norm_kernel, dim3((unsigned)layout->n_utts), dim3(256), 0, (hipStream_t)stream, feats, feats_ld, ncol,
                     layout->d_frame_off);
  HIP_TRY(hipGetLastError());
  return MFCC_OK;
}

int mfcc_pad_batch(const mfcc_layout* layout, const float* feats, int64_t feats_ld, int32_t ncol, int32_t time_steps, float value,
                   int32_t pre_pad, int32_t pre_trunc, float* out, void* stream) {
  if (!layout) return fail(MFCC_ERR_INVALID, "mfcc_pad_batch: null layout");
  if (ncol < 1 || feats_ld < ncol || time_steps < 1) return fail(MFCC_ERR_INVALID, "mfcc_pad_batch: bad shape");
  if ((int64_t)time_steps * ncol > INT32_MAX) return fail(MFCC_ERR_UNSUPPORTED, "mfcc_pad_batch: time_steps * ncol exceeds 2^31-1");
  if (layout->n_utts == 0) return MFCC_OK;
  if (layout->n_utts > 65535) return fail(MFCC_ERR_UNSUPPORTED, "mfcc_pad_batch: more than 65535 utterances per call");
  if (!out || (!feats && layout->total_frames > 0)) return fail(MFCC_ERR_INVALID, "mfcc_pad_batch: null pointer");
  const int total = time_steps * ncol;
  const dim3 grid((unsigned)std::max(1, std::min(64, (total + 255) / 256)), (unsigned)layout->n_utts);
  hipLaunchKernelGGL(mfcc::pad_batch_kernel, grid, dim3(256), 0, (hipStream_t)stream, feats, feats_ld, ncol, layout->d_frame_off, time_steps,
                     value, pre_pad, pre_trunc, out);
  HIP_TRY(hipGetLastError());
  return MFCC_OK;
}

/* ---- streaming ------------------------------------------------------------------------- */

int mfcc_stream_create(const mfcc_plan* plan, int32_t n_streams, int32_t max_chunk, int32_t audio_dtype, mfcc_stream** out) {
  if (!plan || !out) return fail(MFCC_ERR_INVALID, "mfcc_stream_create: null argument");
  *out = nullptr;
  if (n_streams < 1 || n_streams > 65535) return fail(MFCC_ERR_INVALID, "n_streams=%d outside 1..65535", n_streams);
  if (max_chunk < 1 || max_chunk > (1 << 24)) return fail(MFCC_ERR_INVALID, "max_chunk=%d outside 1..2^24", max_chunk);
  if (audio_dtype != MFCC_AUDIO_I16 && audio_dtype != MFCC_AUDIO_F32) return fail(MFCC_ERR_INVALID, "audio_dtype=%d", audio_dtype);
  if (plan->cfg.pad_fac != 2) return fail(MFCC_ERR_UNSUPPORTED, "streaming implements pad_fac=2");
  if (plan->cfg.numcep > mfcc::kMaxCep) return fail(MFCC_ERR_UNSUPPORTED, "streaming keeps at most %d cepstra per row", mfcc::kMaxCep);
  // the emit kernel indexes (stream, row, coefficient) with 32 bits
  if ((int64_t)n_streams * (max_chunk / plan->frame_step + 2 + 2 * mfcc::kHalo) * mfcc::kMaxCep >= ((int64_t)1 << 31))
    return fail(MFCC_ERR_UNSUPPORTED, "n_streams * rows per push = %d * %d exceeds what one emit launch indexes; use fewer streams "
                "or smaller chunks", n_streams, max_chunk / plan->frame_step + 2 + 2 * mfcc::kHalo);
  if (plan->cfg.frame_rule != 0) return fail(MFCC_ERR_UNSUPPORTED, "streaming implements frame_rule=0 (zero-padded last frame)");
  HIP_TRY(hipSetDevice(plan->device));
  mfcc_stream* s = new (std::nothrow) mfcc_stream();
  if (!s) return fail(MFCC_ERR_NOMEM, "mfcc_stream_create: out of host memory");
  s->plan = plan;
  s->device = plan->device;
  s->n_streams = n_streams;
  s->max_chunk = max_chunk;
  s->dtype = audio_dtype;
  // a segment holds the carried tail (< frame_size + frame_step samples) and one chunk
  s->cap = (plan->frame_size + plan->frame_step + max_chunk + 15) / 8 * 8;
  const int32_t max_new = max_chunk / plan->frame_step + 2;
  s->rows = mfcc::kStreamHalo + max_new;
  s->tile_max = mfcc::tile_frames_for(plan->cfg.numcep);
  s->max_tiles = (max_new + s->tile_max - 1) / s->tile_max;
  const size_t esz = audio_dtype == MFCC_AUDIO_I16 ? 2 : 4;
  const size_t seg_bytes = (size_t)2 * n_streams * s->cap * esz;
  const size_t stat_bytes = (size_t)2 * n_streams * s->rows * mfcc::kMaxCep * sizeof(float);
  hipError_t e = hipMalloc(&s->seg, seg_bytes);
  if (e == hipSuccess) e = hipMalloc((void**)&s->stat, stat_bytes);
  if (e == hipSuccess) e = hipMalloc((void**)&s->d_sample_off, (n_streams + 1) * sizeof(int64_t));
  if (e == hipSuccess) e = hipMalloc((void**)&s->d_sample_len, (n_streams + 1) * sizeof(int64_t));
  if (e == hipSuccess) e = hipMalloc((void**)&s->d_frame_off, (n_streams + 1) * sizeof(int64_t));
  if (e == hipSuccess) e = hipMalloc((void**)&s->d_tiles, (size_t)n_streams * s->max_tiles * sizeof(mfcc::Tile));
  if (e == hipSuccess) e = hipMemset(s->seg, 0, seg_bytes);
  if (e == hipSuccess) e = hipMemset(s->stat, 0, stat_bytes);
  if (e != hipSuccess) {
    mfcc_stream_destroy(s);
    return fail(MFCC_ERR_NOMEM, "mfcc_stream_create: %s", hipGetErrorString(e));
  }
  *out = s;
  return MFCC_OK;
}

int mfcc_stream_destroy(mfcc_stream* s) {
  if (!s) return MFCC_OK;
  (void)hipFree(s->seg);
  (void)hipFree(s->stat);
  (void)hipFree(s->d_sample_off);
  (void)hipFree(s->d_sample_len);
  (void)hipFree(s->d_frame_off);
  (void)hipFree(s->d_tiles);
  delete s;
  return MFCC_OK;
}

int mfcc_stream_reset(mfcc_stream* s) {
  if (!s) return fail(MFCC_ERR_INVALID, "mfcc_stream_reset: null stream");
  s->total = s->seg_start = s->f_done = s->e_done = 0;
  s->prev_new_rows = 0;
  return MFCC_OK;
}

int32_t mfcc_stream_max_rows(const mfcc_stream* s) {
  if (!s) return fail(MFCC_ERR_INVALID, "mfcc_stream_max_rows: null stream");
  return s->max_chunk / s->plan->frame_step + 2 + 2 * mfcc::kHalo;
}

// One step: append n samples per stream (n may be 0), compute the static rows of the frames
// f_done .. f_target-1, emit output rows e_done .. e_target-1.
static int stream_step(mfcc_stream* s, const void* chunk, int64_t chunk_ld, int32_t n, int64_t f_target, int64_t e_target,
                       bool last, float* out, int64_t out_stream_stride, int64_t out_ld, hipStream_t st) {
  const mfcc_plan* plan = s->plan;
  const int32_t step = plan->frame_step;
  const int64_t start = s->f_done > 0 ? (s->f_done - 1) * step : 0;  // one frame of history: the pre-emphasis predecessor
  mfcc::StreamStep ss;
  ss.n_streams = s->n_streams;
  ss.cap = s->cap;
  ss.rows = s->rows;
  ss.src_half = s->half;
  ss.src_skip = (int32_t)(start - s->seg_start);
  ss.keep = (int32_t)(s->total - start);
  ss.n_new = n;
  ss.prev_new_rows = s->prev_new_rows;
  ss.t0 = s->f_done > 0 ? 1 : 0;
  ss.nt = (int32_t)(f_target - s->f_done);
  ss.tile_max = s->tile_max;
  ss.tiles_per_stream = (ss.nt + s->tile_max - 1) / s->tile_max;
  if (ss.keep + n > s->cap || ss.nt > s->rows - mfcc::kStreamHalo || ss.tiles_per_stream > s->max_tiles)
    return fail(MFCC_ERR_INVALID, "mfcc_stream: internal capacity exceeded (keep %d, n %d, new frames %d)", ss.keep, n, ss.nt);
  const int64_t k_rows = e_target - s->e_done;
  // frame numbers of the rows to emit, relative to the first new static row (utterance frame f_done)
  const int64_t big = 1 << 28;
  const int32_t e0 = (int32_t)(s->e_done - s->f_done);
  const int32_t lo = (int32_t)std::max(-s->f_done, -big);
  const int32_t hi = last ? (int32_t)(f_target - 1 - s->f_done) : (int32_t)big;
  const int32_t new_half = s->half ^ 1;
  auto advance = [&]() {
    s->half = new_half;
    s->seg_start = start;
    s->total += n;
    s->prev_new_rows = ss.nt;
    s->f_done = f_target;
    s->e_done = e_target;
  };
  if (!plan->general && !plan->plain_only && ss.tiles_per_stream <= 1) {
    // the usual push (new frames fit one tile): one launch, workgroup b = stream b, does all three parts
    mfcc::FusedParams fp;
    fp.audio = s->seg;
    fp.total_elems = (int64_t)2 * s->n_streams * s->cap;
    fp.sample_off = nullptr;
    fp.sample_len = nullptr;
    fp.frame_off = nullptr;
    fp.tiles = nullptr;
    fp.out = s->stat;
    fp.out_ld = mfcc::kMaxCep;
    fp.pc = plan->pc;
    fp.stack_n = 0;
    fp.tile_sums = nullptr;
    fp.wg_tiles = nullptr;
    fp.wg_off = nullptr;
#ifdef MFCC_DIAG
    fp.dbg = nullptr;
#endif
    fp.stream.step = ss;
    fp.stream.chunk = chunk;
    fp.stream.chunk_ld = chunk_ld;
    fp.stream.rows_out = out;
    fp.stream.out_stream_stride = out_stream_stride;
    fp.stream.out_ld = out_ld;
    fp.stream.e0 = e0;
    fp.stream.k = (int32_t)std::max<int64_t>(k_rows, 0);
    fp.stream.lo = lo;
    fp.stream.hi = hi;
    launch_fused(plan, fp, s->n_streams, kLaunchStream, s->dtype == MFCC_AUDIO_I16, st);
    HIP_TRY(hipGetLastError());
    advance();
    return MFCC_OK;
  }
  const int total = ss.keep + n;
  const dim3 agrid((unsigned)std::max(1, std::min(64, (total + 255) / 256)), (unsigned)s->n_streams);
  if (s->dtype == MFCC_AUDIO_I16)
    hipLaunchKernelGGL(mfcc::stream_append_kernel<int16_t>, agrid, dim3(256), 0, st, ss, (int16_t*)s->seg, s->stat,
                       (const int16_t*)chunk, chunk_ld, s->d_sample_off, s->d_sample_len, s->d_frame_off, s->d_tiles);
  else
    hipLaunchKernelGGL(mfcc::stream_append_kernel<float>, agrid, dim3(256), 0, st, ss, (float*)s->seg, s->stat,
                       (const float*)chunk, chunk_ld, s->d_sample_off, s->d_sample_len, s->d_frame_off, s->d_tiles);
  if (ss.nt > 0) {
    mfcc::FusedParams fp;
    fp.audio = s->seg;
    fp.total_elems = (int64_t)2 * s->n_streams * s->cap;
    fp.sample_off = s->d_sample_off;
    fp.sample_len = s->d_sample_len;
    fp.frame_off = s->d_frame_off;
    fp.tiles = s->d_tiles;
    fp.out = s->stat;
    fp.out_ld = mfcc::kMaxCep;
    fp.pc = plan->pc;
    fp.stack_n = 0;
    fp.tile_sums = nullptr;
    fp.wg_tiles = nullptr;
    fp.wg_off = nullptr;
#ifdef MFCC_DIAG
    fp.dbg = nullptr;
#endif
    launch_fused(plan, fp, (int64_t)s->n_streams * ss.tiles_per_stream, MFCC_OUT_STATIC, s->dtype == MFCC_AUDIO_I16, st);
  }
  const int64_t k = k_rows;
  if (k > 0) {
    const int64_t work = (int64_t)s->n_streams * k * mfcc::kMaxCep;
    hipLaunchKernelGGL(mfcc::stream_emit_kernel, dim3(blocks_for(work, 256)), dim3(256), 0, st, s->stat, new_half, s->n_streams,
                       s->rows, e0, (int32_t)k, lo, hi, plan->cfg.numcep, out, out_stream_stride, out_ld);
  }
  HIP_TRY(hipGetLastError());
  advance();
  return MFCC_OK;
}

int mfcc_stream_push(mfcc_stream* s, const void* chunk, int64_t chunk_ld, int32_t n, float* out, int64_t out_stream_stride,
                     int64_t out_ld, int32_t* n_rows_host, void* stream) {
  if (!s) return fail(MFCC_ERR_INVALID, "mfcc_stream_push: null stream");
  if (n_rows_host) *n_rows_host = 0;
  if (n < 0 || n > s->max_chunk) return fail(MFCC_ERR_INVALID, "chunk of %d samples, the stream was created for at most %d", n, s->max_chunk);
  if (n == 0) return MFCC_OK;
  if (!chunk || chunk_ld < n) return fail(MFCC_ERR_INVALID, "mfcc_stream_push: bad chunk pointer or stride");
  const int32_t width = 3 * s->plan->cfg.numcep;
  const int64_t total = s->total + n;
  const int64_t complete = total >= s->plan->frame_size ? (total - s->plan->frame_size) / s->plan->frame_step + 1 : 0;
  const int64_t emit = std::max<int64_t>(complete - mfcc::kHalo, s->e_done);  // rows whose +4-frame context exists
  const int64_t k = emit - s->e_done;
  if (k > 0 && (!out || out_ld < width || out_stream_stride < k * out_ld))
    return fail(MFCC_ERR_INVALID, "mfcc_stream_push: out must hold %lld rows of %d floats per stream", (long long)k, width);
  const int rc = stream_step(s, chunk, chunk_ld, n, complete, emit, false, out, out_stream_stride, out_ld, (hipStream_t)stream);
  if (rc) return rc;
  if (n_rows_host) *n_rows_host = (int32_t)k;
  return MFCC_OK;
}

int mfcc_stream_flush(mfcc_stream* s, float* out, int64_t out_stream_stride, int64_t out_ld, int32_t* n_rows_host, void* stream) {
  if (!s) return fail(MFCC_ERR_INVALID, "mfcc_stream_flush: null stream");
  if (n_rows_host) *n_rows_host = 0;
  if (s->total == 0) return MFCC_OK;  // nothing was pushed: no utterance, no rows
  const int32_t width = 3 * s->plan->cfg.numcep;
  const int64_t frames = mfcc::frames_for_length(s->total, s->plan->frame_size, s->plan->frame_step);
  const int64_t k = frames - s->e_done;
  if (!out || out_ld < width || out_stream_stride < k * out_ld)
    return fail(MFCC_ERR_INVALID, "mfcc_stream_flush: out must hold %lld rows of %d floats per stream", (long long)k, width);
  const int rc = stream_step(s, nullptr, 0, 0, frames, frames, true, out, out_stream_stride, out_ld, (hipStream_t)stream);
  if (rc) return rc;
  if (n_rows_host) *n_rows_host = (int32_t)k;
  return mfcc_stream_reset(s);
}

}  // extern "C"
