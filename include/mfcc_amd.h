/*
 * mfcc_amd.h -- C ABI of the MI355X (gfx950) MFCC front-end.
 *
 * The reference (Robomate/mfcc) has no FFI layer: its hot path is thirteen
 * module-level numpy functions in mfcc-self5.py / mfcc-self15.py called only by
 * their own main().  Each entry point below therefore cites the reference
 * FUNCTION it replaces (file:line of the reference repository); the ctypes
 * binding a maintainer adds is shipped as mfcc_amd/_capi.py.
 *
 * Conventions
 *   - every function returns MFCC_OK (0) or a negative MFCC_ERR_* code; the
 *     text of the last error on the calling thread is mfcc_last_error();
 *   - pointers are DEVICE pointers unless the parameter name ends in _host;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); all
 *     device work is enqueued on it and the call returns without waiting;
 *   - a plan / layout belongs to the device it was created on; the caller
 *     keeps that device current while using it;
 *   - no call allocates device memory except the *_create functions and the first
 *     mfcc_cmvn_accumulate on a (device, stream) pair (16 MiB, kept until mfcc_cmvn_release).
 */
#ifndef MFCC_AMD_H
#define MFCC_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MFCC_OK 0
#define MFCC_ERR_INVALID (-1)     /* bad argument                                   */
#define MFCC_ERR_UNSUPPORTED (-2) /* configuration outside what the kernels cover   */
#define MFCC_ERR_HIP (-3)         /* a HIP runtime call failed                      */
#define MFCC_ERR_NOMEM (-4)       /* host or device allocation failed               */

/* audio sample types (mfcc-self5.py:32-36 loadrawdata -> int16; wav.read :196) */
#define MFCC_AUDIO_I16 0
#define MFCC_AUDIO_F32 1

/* which spectrum feeds the mel bank: mfcc-self5.py:217 (power) / mfcc-self15.py:234 (magnitude) */
#define MFCC_SPECTRUM_POWER 0
#define MFCC_SPECTRUM_MAG 1

/* mfcc-self5.py:78,95: np.hamming is computed but NOT applied -> NONE is the reference behaviour */
#define MFCC_WINDOW_NONE 0
#define MFCC_WINDOW_HAMMING 1

/* Which kernel a plan runs.  AUTO takes the kernel specialised for the reference's own framings
 * (400/160 samples at 16 kHz with the 512-point FFT; 200/80 at 8 kHz with the 256-point FFT winfft
 * derives or an explicit 512-point one; also 320/160 samples, frame_time = 20 at 16 kHz, for every
 * entry point but mfcc_forward_stacked; pad_fac 2, <= 40 filters, <= 16 cepstra) when the
 * configuration fits it and the general one otherwise, which covers every other
 * (fs, frame_time, step_time) winfft accepts (mfcc-self5.py:75-84: frame sizes up to 2048 samples,
 * FFT lengths 2 .. 2048) and any pad_fac of delta (:169).  GENERAL forces the general kernel. */
#define MFCC_KERNEL_AUTO 0
#define MFCC_KERNEL_GENERAL 1

/* what mfcc_forward writes, one row per frame */
#define MFCC_OUT_MFCC39 0  /* [F, 3*numcep]  c | delta | delta-delta   (mfcc-self5.py:181-186) */
#define MFCC_OUT_STATIC 1  /* [F, numcep]    c0 := ln E               (mfcc-self5.py:158-167) */
#define MFCC_OUT_LOGMEL 2  /* [F, num_bins+1] ln mel | ln E           (mfcc-self5.py:119-134) */
#define MFCC_OUT_SPEC 3    /* [F, nfft/2+1]  |X|^2/nfft or |X|        (mfcc-self5.py:96-99)  */
#define MFCC_OUT_LOGSPEC 4 /* [F, nfft/2+1]  natural log of the above                        */

/* Mirrors the reference's module-level globals (mfcc-self5.py:21-29) and the
 * keyword defaults of its functions.  mfcc_config_default() fills the values of
 * mfcc-self5.py's main(). */
typedef struct mfcc_config {
  double fs;          /* 16000            mfcc-self5.py:28                      */
  double frame_time;  /* 25 (ms)          mfcc-self5.py:21, winfft :70,75       */
  double step_time;   /* 10 (ms)          mfcc-self5.py:22, winfft :70,76       */
  int32_t num_bins;   /* 26               mfcc-self5.py:23, melbanks :102       */
  double low_freq;    /* 0                mfcc-self5.py:24, melbanks :105       */
  int32_t numcep;     /* 13               mfcc-self5.py:219 (staticfeatures hard-codes 13, :161) */
  double lift_factor; /* 22; <=0 = off    mfcc-self5.py:26, dctopt :144-156     */
  int32_t pad_fac;    /* 2                delta :169                            */
  int32_t use_preemph;/* 1 (self5) / 0 (self15, mfcc-self15.py:230)             */
  double preemph;     /* 0.97             preemphasis :66                       */
  int32_t spectrum;   /* MFCC_SPECTRUM_*                                        */
  int32_t window;     /* MFCC_WINDOW_*                                          */
  int32_t floor_mel;  /* 0 = ln of the unfloored mel sum, -inf on silence (mfcc-self5.py:126-127);
                         1 = floor exact zeros to float64 eps first            */
  int32_t num_fft;    /* 0 = next power of two >= frame size as winfft derives it (mfcc-self5.py:79-84);
                         otherwise the explicit FFT length of the callers that pass nfft=512
                         (inference_ctc_eri_05.py:253-257)                     */
  int32_t frame_rule; /* 0 = zero-pad the last frame: F = 1 + ceil(max(0, L - size) / step)  (winfft, mfcc-self5.py:85-92);
                         1 = complete frames only: F = max(0, fix((L - (size - step)) / step))
                             (spec_anal, anal_recog_nn_liste.m:503)                     */
  double log_floor;   /* MFCC_OUT_LOGSPEC takes ln(max(v, log_floor)) when > 0: the log(max(|X|, 1)) of
                         spec_anal, anal_recog_nn_liste.m:529; 0 = plain ln            */
  double high_freq;   /* 0 = fs/2, the upper edge melbanks always uses (mfcc-self5.py:103); otherwise the
                         upper edge in Hz (the highfreq argument of the CTC scripts' MFCC routine) */
  int32_t kernel;     /* MFCC_KERNEL_*                                             */
  int32_t keep_c0;    /* 0 = coefficient 0 is replaced by ln E as staticfeatures does (mfcc-self5.py:164);
                         1 = keep the DCT's own c0 (appendEnergy=False of the CTC scripts' MFCC routine;
                             runs the general kernel)                                          */
} mfcc_config;

typedef struct mfcc_plan mfcc_plan;     /* immutable constant tables on one device          */
typedef struct mfcc_layout mfcc_layout; /* geometry of one ragged batch (offsets, tile list) */
typedef struct mfcc_stream mfcc_stream; /* state of a batch of live audio streams           */

/* ---- library ------------------------------------------------------------- */
int mfcc_version(void);
const char* mfcc_last_error(void);
void mfcc_config_default(mfcc_config* cfg); /* = mfcc-self5.py main(), :211-229 */
int32_t mfcc_config_size(void);             /* sizeof(mfcc_config) of this build: lets a binding check its struct */

/* ---- host-side constants (double precision, no device needed) -------------- */
/* melbanks(fs, numbins, lowfreq, num_fft) -> [numbins, num_fft/2+1] row-major.  mfcc-self5.py:102-117 */
int mfcc_melbank_host(double fs, int32_t numbins, double lowfreq, int32_t num_fft, double* out_host);
/* frame_size, frame_step, num_fft exactly as winfft derives them.  mfcc-self5.py:75-84 */
int mfcc_frame_geometry(double fs, double frame_time, double step_time, int32_t* frame_size, int32_t* frame_step,
                        int32_t* num_fft);
/* pos+1 of winfft's padding loop.  mfcc-self5.py:85-92 */
int64_t mfcc_frames_for_length(int64_t n_samples, int32_t frame_size, int32_t frame_step);

/* ---- plan ---------------------------------------------------------------- */
int mfcc_plan_create(const mfcc_config* cfg, int device, mfcc_plan** out);
int mfcc_plan_destroy(mfcc_plan* plan);
int mfcc_plan_info(const mfcc_plan* plan, int32_t* frame_size, int32_t* frame_step, int32_t* num_fft,
                   int32_t* num_bins, int32_t* numcep);
/* MFCC_KERNEL_GENERAL if the plan runs the general kernel, MFCC_KERNEL_AUTO (0) if the specialised one */
int32_t mfcc_plan_kernel(const mfcc_plan* plan);
/* 1 if mfcc_forward serves out_kind on this plan, 0 if not.  Every plan serves every kind with one
 * exception: a general-kernel plan whose FFT buffers and filter rows leave no room in a workgroup's
 * 64 KiB of LDS for the static and delta rows of a tile (2 * numcep floats for each of 4 + 4 * pad_fac
 * rows; e.g. num_fft 2048 with 128 filters, 64 cepstra and pad_fac 9) has no fused MFCC_OUT_MFCC39: take
 * MFCC_OUT_STATIC and apply mfcc_delta (delta, mfcc-self5.py:169-179) per utterance. */
int32_t mfcc_plan_supports(const mfcc_plan* plan, int32_t out_kind);

/* ---- batch layout ---------------------------------------------------------- */
/* n_utts utterances stored back to back: utterance u occupies samples
 * [sample_offset[u], sample_offset[u+1]) of `audio` and rows
 * [frame_offset[u], frame_offset[u+1]) of `out`.  The reference processes one
 * utterance per main() call (mfcc-self5.py:188-229); every per-utterance rule
 * (y[0]=x[0], tail zero padding, delta edge replication) is applied per entry. */
int mfcc_layout_create(const mfcc_plan* plan, const int64_t* lengths_host, int64_t n_utts, mfcc_layout** out);
int mfcc_layout_destroy(mfcc_layout* layout);
int mfcc_layout_info(const mfcc_layout* layout, int64_t* n_utts, int64_t* total_samples, int64_t* total_frames);
/* The same container for feature rows that did not come from audio through this library
 * (.pat files, MLP_5hidlayer_40.py:61-68): utterance u owns frame_counts[u] >= 0 consecutive rows.
 * Accepted by the row-wise entry points (stacking, padding, per-utterance normalisation), not by
 * mfcc_forward / mfcc_fir. */
int mfcc_layout_create_frames(int device, const int64_t* frame_counts_host, int64_t n_utts, mfcc_layout** out);
/* copies the (n_utts+1) prefix sums to host arrays; either pointer may be NULL */
int mfcc_layout_offsets_host(const mfcc_layout* layout, int64_t* sample_offsets_host, int64_t* frame_offsets_host);

/* ---- the fused hot path ------------------------------------------------------
 * audio -> pre-emphasis -> framing -> rFFT -> spectrum -> mel -> ln -> DCT -> lifter
 *       -> c0:=lnE -> delta -> delta-delta -> out rows.
 * Replaces main()'s call chain mfcc-self5.py:211-229 (mfcc-self15.py:228-246).
 * `out` has `out_ld` floats per row (>= the row width of out_kind; plans of the specialised
 * kernel take out_ld up to 6 391 320 floats, MFCC_ERR_UNSUPPORTED beyond: its row offsets inside
 * a tile are 32-bit). */
int mfcc_forward(const mfcc_plan* plan, const mfcc_layout* layout, const void* audio, int32_t audio_dtype,
                 float* out, int64_t out_ld, int32_t out_kind, void* stream);
int32_t mfcc_out_width(const mfcc_plan* plan, int32_t out_kind);

/* mfcc_forward(MFCC_OUT_MFCC39) that collects, in the same launch, the statistics of a corpus normalisation over
 * the rows it writes -- the constants mm, vv of (x + mm) .* vv, anal_recog_nn_liste.m:180, come from them -- so that
 * no second pass reads the rows back: sums (float64[1 + 6 numcep] on the device, the layout mfcc_cmvn_accumulate
 * keeps) gets sums[0] += total_frames, sums[1 + c] += sum of column c, sums[1 + 3 numcep + c] += sum of its squares.
 * The rows are bit-identical to mfcc_forward's.  The writing lanes add their values to float32 column sums of their
 * tile (at most 296 rows) in LDS; a second small kernel adds the tiles in float64.  The order of every addition is
 * fixed by the code, so equal inputs give equal bits; against mfcc_cmvn_accumulate over the same rows (float64
 * throughout) the sums differ by about 1e-7 relative.  workspace: device memory of mfcc_forward_sums_workspace()
 * bytes, 4-byte aligned, written and read by this call only (calls that overlap in time need one each).
 * Plans of the specialised kernel (mfcc_plan_kernel() == MFCC_KERNEL_AUTO); MFCC_ERR_UNSUPPORTED otherwise. */
int64_t mfcc_forward_sums_workspace(const mfcc_plan* plan, const mfcc_layout* layout);
int mfcc_forward_sums(const mfcc_plan* plan, const mfcc_layout* layout, const void* audio, int32_t audio_dtype,
                      float* out, int64_t out_ld, double* sums, void* workspace, void* stream);

/* The "257 coefficients x N frames" input of the MLP / CNN scripts (coeffs = 257, framenum = 17,
 * MLP_5hidlayer_40.py:244-248) straight from audio in one launch: winfft's spectrum rows
 * (mfcc-self5.py:96-99; their natural log for MFCC_OUT_LOGSPEC) stacked as padUtterance + slidingWin do
 * (MLP_5hidlayer_40.py:70-105): out[t, f*B + c] = spec[clamp(t + f - (N-1)/2, 0, F_u - 1), c], B = nfft/2+1.
 * Every spectrum row is computed once, kept in LDS, and each output row leaves as one contiguous write.
 * out: [total_frames, framenum * B] contiguous floats, 16-byte aligned; out_kind MFCC_OUT_SPEC or
 * MFCC_OUT_LOGSPEC; framenum odd, 1 .. 21 (the reference's scripts use 7 .. 21, MLP_5hidlayer_40.py:244-262;
 * MFCC_ERR_UNSUPPORTED beyond: use mfcc_forward + mfcc_stack_context, which take any odd framenum and any
 * feature width).  Windows of 19 and 21 frames keep more spectrum rows in LDS and run one workgroup per CU.
 * out_kind MFCC_OUT_MFCC39 stacks the finished feature rows instead (coeffs = 39, e.g. framenum = 11, the
 * reference's best model, nn_model_statistics.csv:13): out[t, f*D + j] = mfcc39[clamp(t + f - (N-1)/2, 0, F_u - 1), j],
 * D = 3*numcep, out [total_frames, framenum * D]; framenum odd, 3 .. 21; plans of the specialised kernel. */
int mfcc_forward_stacked(const mfcc_plan* plan, const mfcc_layout* layout, const void* audio, int32_t audio_dtype,
                         int32_t framenum, float* out, int32_t out_kind, void* stream);

/* ---- the reference's stages one at a time (float32 device arrays) ------------- */
/* preemphasis(signal, coeff).  mfcc-self5.py:66-68 */
int mfcc_preemphasis(const void* signal, int32_t dtype, int64_t n, double coeff, float* out, void* stream);
/* melspec(spec_list, melbank): ln(spec @ melbank^T) [F,nmel] and floored ln(sum spec) [F].  mfcc-self5.py:119-134 */
int mfcc_melspec(const float* spec, int64_t n_frames, int32_t n_bins, const float* melbank, int32_t nmel,
                 int32_t floor_mel, float* logmel, float* loge, void* stream);
/* mfcc(melspec_list, numcep): DCT-II 'ortho', first numcep.  mfcc-self5.py:136-142 */
int mfcc_dct(const float* logmel, int64_t n_frames, int32_t nmel, int32_t numcep, float* out, void* stream);
/* dctopt(mfcc_list, lift_factor).  mfcc-self5.py:144-156 */
int mfcc_lifter(const float* cep, int64_t n_frames, int32_t ncoeff, double lift_factor, float* out, void* stream);
/* delta(features, pad_fac) over ONE utterance of n_frames rows.  mfcc-self5.py:169-179 */
int mfcc_delta(const float* feats, int64_t n_frames, int32_t ncol, int32_t pad_fac, float* out, void* stream);
/* strided column copy dst[r, 0:ncol] = src[r, 0:ncol] -- staticfeatures' c0 overwrite (:164) and
 * spectrograms' concatenation (:184-185) are built from it */
int mfcc_copy_cols(const float* src, int64_t src_ld, float* dst, int64_t dst_ld, int64_t n_rows, int32_t ncol,
                   void* stream);

/* Centred FIR filter per utterance of the layout: out[n] = sum_k taps[k] * x[n + delay - k],
 * samples outside the utterance count as zero.  With the 15 taps of initdft.bpre and delay 7
 * this is spec_anal's pre-emphasis, filter() followed by the group-delay shift
 * (anal_recog_nn_liste.m:56-57, :511-516).  taps_host: at most 32 coefficients, read at call time. */
int mfcc_fir(const mfcc_layout* layout, const void* audio, int32_t audio_dtype, const float* taps_host, int32_t ntaps,
             int32_t delay, float* out, void* stream);

/* ---- context stacking ----------------------------------------------------------
 * padUtterance + slidingWin per utterance of the layout; N = framenum (odd), h = (N-1)/2,
 * feats is [total_frames, C] with row stride feats_ld.  Output row i holds the window of row
 * t = row_index ? row_index[i] : i (row_index: device int64[n_out], rows of the packed array;
 * NULL = all rows in order, n_out ignored) in one of two element orders:
 *   MFCC_STACK_FRAME_MAJOR  out[i, f*C + c] = feats[clamp(t + f - h), c]   the MLP vector
 *                           (reshape(..., order='F') of the [C,N] slice, MLP_5hidlayer_40.py:70-105)
 *   MFCC_STACK_TILE         out[i, c*N + f] = feats[clamp(t + f - h), c]   the [C,N] slice itself, as the
 *                           CNN scripts keep it (CNN_DenseNet40_6.py:98,124-131)
 * row_index carries the shuffled window order of CNN_DenseNet40_6.py:91-97 (createRandomvec). */
#define MFCC_STACK_FRAME_MAJOR 0
#define MFCC_STACK_TILE 1
int mfcc_stack_context(const mfcc_layout* layout, const float* feats, int64_t feats_ld, int32_t ncol,
                       int32_t framenum, int32_t out_layout, const int64_t* row_index, int64_t n_out, float* out,
                       void* stream);

/* ---- CMVN ---------------------------------------------------------------------------
 * sums[0] += n_rows, sums[1..D] += sum_x, sums[1+D..2D] += sum_x^2 (float64, device).
 * Generalises the per-utterance (x-mean)/std of inference_ctc_eri_05.py:291 and the
 * corpus constants (x+mm).*vv of anal_recog_nn_liste.m:180.  The caller zeroes `sums`
 * and combines it across ranks.  The reduction order is fixed by the code (per-workgroup
 * partial sums, then a fixed-order second stage; no atomics), so equal inputs give equal bits.
 * The partial sums live in a 16 MiB buffer that the library allocates at the first call on a
 * (device, stream) pair -- sized for the widest rows, so no later call allocates, frees or waits --
 * and keeps until mfcc_cmvn_release; ncol <= 1024.  That first call may not sit inside a stream
 * capture (it allocates): callers that capture, or that create and destroy streams freely, use
 * mfcc_cmvn_accumulate_static below. */
int mfcc_cmvn_accumulate(const float* feats, int64_t feats_ld, int64_t n_rows, int32_t ncol, double* sums,
                         void* stream);
/* Frees every buffer mfcc_cmvn_accumulate holds on the current device.  The caller makes sure that
 * no such call is still running there (e.g. after hipDeviceSynchronize). */
int mfcc_cmvn_release(void);
/* No counterpart in the reference.  The same operation with the same results for callers that
 * cannot allocate device memory at call time (code under HIP graph capture, serving loops,
 * harnesses that forbid hipMalloc): the partial sums go to a fixed 4.3 MB device array that is
 * part of the library image (one copy per GPU, held until the process ends), so the call never
 * allocates.  ncol <= MFCC_CMVN_STATIC_MAX_COLS, otherwise MFCC_ERR_UNSUPPORTED.  All calls on one
 * GPU share that array.  Eager calls take turns on it whatever their streams: each waits (on its
 * stream, not on the host) for an event the call before it on that GPU recorded.  A call under
 * stream capture issues neither the wait nor the record: launches of a graph that contains it, and
 * eager calls beside them, race on the array unless the caller orders them. */
#define MFCC_CMVN_STATIC_MAX_COLS 264
int mfcc_cmvn_accumulate_static(const float* feats, int64_t feats_ld, int64_t n_rows, int32_t ncol, double* sums,
                                void* stream);
/* feats[r,c] = (feats[r,c] + mm[c]) * vv[c] in place; mm, vv are float32[ncol] on the device */
int mfcc_cmvn_apply(float* feats, int64_t feats_ld, int64_t n_rows, int32_t ncol, const float* mm, const float* vv,
                    void* stream);

/* ---- sequence shaping for the recurrent / CTC consumers ------------------------------------
 * In place per utterance of the layout: (x - mean) / std with mean and population std over ALL
 * frames x coefficients of the utterance (inference_ctc_eri_05.py:291, CTC/ctc_test_1.py:54). */
int mfcc_utt_scalar_norm(const mfcc_layout* layout, float* feats, int64_t feats_ld, int32_t ncol, void* stream);
/* pad_sequences(sequences, maxlen, padding, truncating, value), CTC/utils.py:71-128 (and the
 * zero-padded [batch, time_steps, coeffs] arrays of BiLSTM_7.py:82-124): ragged rows of the
 * layout -> out[n_utts, time_steps, ncol]; pre_pad / pre_trunc = 1 select 'pre'. */
int mfcc_pad_batch(const mfcc_layout* layout, const float* feats, int64_t feats_ld, int32_t ncol, int32_t time_steps,
                   float value, int32_t pre_pad, int32_t pre_trunc, float* out, void* stream);

/* ---- streaming ------------------------------------------------------------------------
 * n_streams live streams that advance in lock step (every push brings the same number of
 * samples for each).  The reference has no incremental routine -- its live script queues
 * microphone blocks (get_mic_input_13.py:28-65) and featurises a whole recording
 * (inference_ctc_eri_05.py:464-487) -- so the contract is: the rows returned by all pushes and
 * the final flush, concatenated, equal mfcc_forward(MFCC_OUT_MFCC39) of the concatenated audio
 * (main(), mfcc-self5.py:211-229), including y[0]=x[0], the zero-padded last frame and the
 * delta edge replication.  A row is returned once the 4 frames after it are complete (delta of
 * delta reaches 4 frames ahead); flush returns the rest and starts a new utterance.
 *
 * chunk: [n_streams, n] samples of the dtype given at creation, row stride chunk_ld;
 * out:   stream b's rows at out + b*out_stream_stride + i*out_ld, i < *n_rows_host, which is
 *        known on return (host bookkeeping only; the device work is enqueued on `stream`).
 * All calls on one mfcc_stream must use the same HIP stream (or be ordered by the caller).
 * Nothing is allocated after create. */
int mfcc_stream_create(const mfcc_plan* plan, int32_t n_streams, int32_t max_chunk, int32_t audio_dtype,
                       mfcc_stream** out);
int mfcc_stream_destroy(mfcc_stream* s);
int mfcc_stream_reset(mfcc_stream* s); /* drop pending state; the next push starts new utterances */
int32_t mfcc_stream_max_rows(const mfcc_stream* s); /* upper bound of rows per stream of one push / flush */
int mfcc_stream_push(mfcc_stream* s, const void* chunk, int64_t chunk_ld, int32_t n, float* out,
                     int64_t out_stream_stride, int64_t out_ld, int32_t* n_rows_host, void* stream);
int mfcc_stream_flush(mfcc_stream* s, float* out, int64_t out_stream_stride, int64_t out_ld, int32_t* n_rows_host,
                      void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MFCC_AMD_H */
