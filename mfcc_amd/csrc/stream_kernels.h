// Kernels of the streaming path: B live streams advance in lock step; each push appends a
// chunk per stream, the fused kernel (static-feature mode) turns the new complete frames into
// static rows, and stream_emit_kernel forms delta / delta-delta for the rows whose +-4-frame
// context is complete.  The reference has no incremental path (its live script queues
// microphone blocks, get_mic_input_13.py:28-65, and featurises whole utterances,
// inference_ctc_eri_05.py:464-487); results are defined by equality with the whole-utterance
// path (mfcc-self5.py:211-229) on the concatenated audio.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fused_kernel.h"

namespace mfcc {

// One launch per push: (1) new segment = [carried tail | chunk], (2) the last kStreamHalo static
// rows move to the front of the other half, (3) the per-step layout of the fused kernel
// (offsets, lengths, tiles) is written from the scalars above -- no host-to-device copy.
template <typename T>
__global__ void __launch_bounds__(256) stream_append_kernel(const StreamStep s, T* seg, float* stat, const T* chunk,
                                                            int64_t chunk_ld, int64_t* sample_off, int64_t* sample_len,
                                                            int64_t* frame_off, Tile* tiles) {
  const int b = blockIdx.y;
  const int dst_half = s.src_half ^ 1;
  const int total = s.keep + s.n_new;
  stream_carry<T>(s, b, seg, stat, chunk, chunk_ld, blockIdx.x * 256 + threadIdx.x, gridDim.x * 256, blockIdx.x == 0, threadIdx.x);
  if (blockIdx.x == 0) {
    if (threadIdx.x == 0) {
      sample_off[b] = ((int64_t)dst_half * s.n_streams + b) * s.cap;
      sample_len[b] = total;
      // row (frame_off + t) of the statics array holds segment frame t; the first new frame t0
      // lands on row kStreamHalo of this stream's block
      frame_off[b] = ((int64_t)dst_half * s.n_streams + b) * s.rows + kStreamHalo - s.t0;
      for (int j = 0; j < s.tiles_per_stream; ++j) {
        const int first = s.t0 + j * s.tile_max;
        const int n = min(s.tile_max, s.t0 + s.nt - first);
        tiles[b * s.tiles_per_stream + j] = Tile{b, first, n > 0 ? n : 0, 0};
      }
      if (b == 0) {
        sample_off[s.n_streams] = 0;  // never read: lengths come from sample_len
        frame_off[s.n_streams] = 0;
      }
    }
  }
}

// Rows e0 .. e0+k-1 of every stream: [c | delta | delta-delta] with the regression of
// mfcc-self5.py:169-179 applied twice.  Frame numbers are relative to the first frame of this
// step's new statics (which sits at row kStreamHalo of the stream's block) and are clamped to
// [lo, hi]: lo = the utterance's frame 0, hi = its last frame at flush (edge replication,
// mfcc-self5.py:174) or out of reach before.
__global__ void __launch_bounds__(256) stream_emit_kernel(const float* stat, int32_t half, int32_t n_streams, int32_t rows,
                                                          int32_t e0, int32_t k, int32_t lo, int32_t hi, int32_t ncep,
                                                          float* out, int64_t out_stream_stride, int64_t out_ld) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  const int c = idx % kMaxCep;
  const int i = (idx / kMaxCep) % k;
  const int b = idx / (kMaxCep * k);
  if (b >= n_streams || c >= ncep) return;
  const float* const S = stat + (((int64_t)half * n_streams + b) * rows + kStreamHalo) * kMaxCep + c;
  stream_emit_one(S, e0 + i, lo, hi, ncep, c, out + b * out_stream_stride + i * out_ld);
}

}  // namespace mfcc
