"""CPU restatement (numpy, float64) of the reference MFCC front-end.

TEST INFRASTRUCTURE ONLY -- never imported by the product package.

Every function cites the reference lines it follows.  Citations are into
``/root/reference``; ``self5`` = ``mfcc-self5.py`` and ``self15`` =
``mfcc-self15.py`` (the 13 pipeline functions are identical in both files,
only ``main`` differs).

Two execution styles are offered:

* ``loop=True``  -- one Python iteration per frame, exactly the reference's
  control flow.  This is the "port" timed as ``cpu_baseline`` in ``bench.py``.
* ``loop=False`` -- whole-utterance array operations (same arithmetic, batched
  library calls).  Used by the tests for larger inputs; agrees with the loop
  form to ~1e-12.

Parity pinned against golden vectors produced by the reference's own
functions (``oracle/gen_golden.py`` -> ``tests/golden/*.npz``).
"""
from __future__ import annotations

import struct

import numpy as np
from scipy.fftpack import dct as _scipy_dct

EPS64 = float(np.finfo(float).eps)  # self5:126,131 (np.finfo(float).eps)

# Defaults of the reference's module-level globals (self5:21-29).
FS = 16000
FRAME_TIME = 25
STEP_TIME = 10
NUM_BINS = 26
LOW_FREQ = 0
LIFT_FACTOR = 22

VARIANTS = {
    # self5:211-217 -- pre-emphasised signal, power spectrum into the mel bank
    "self5": dict(preemph=0.97, spectrum="power"),
    # self15:228-234 -- raw signal into winfft, magnitude spectrum into the mel bank
    "self15": dict(preemph=None, spectrum="mag"),
}


# --------------------------------------------------------------------------
# framing geometry
# --------------------------------------------------------------------------
def frame_geometry(fs=FS, frame_time=FRAME_TIME, step_time=STEP_TIME):
    """(frame_size, frame_step, num_fft) as derived in self5:75-84."""
    frame_size = int(fs * (frame_time * (1.0 / 1000)))
    frame_step = int(fs * (step_time * (1.0 / 1000)))
    num_fft = 1
    # self5:79-84 starts from 0 and doubles 2**k until >= frame_size; for any
    # frame_size >= 1 this is the next power of two >= frame_size.
    while num_fft < frame_size:
        num_fft *= 2
    return frame_size, frame_step, num_fft


def frame_count(n_samples, frame_size=400, frame_step=160):
    """Number of frames produced by winfft (self5:85-92): smallest pos with
    pos*step + size >= L, frames = pos + 1."""
    n_samples = int(n_samples)
    if n_samples <= frame_size:
        return 1
    return 1 + -(-(n_samples - frame_size) // frame_step)


# --------------------------------------------------------------------------
# stages
# --------------------------------------------------------------------------
def preemphasis(signal, coeff=0.97):
    """y[0]=x[0]; y[n]=x[n]-coeff*x[n-1]  (self5:66-68)."""
    signal = np.asarray(signal)
    out = np.empty(signal.shape[0], dtype=np.float64)
    out[0] = signal[0]
    out[1:] = signal[1:] - coeff * signal[:-1]
    return out


def winfft(audio, fs=FS, frame_time=FRAME_TIME, step_time=STEP_TIME, loop=True, num_fft=None):
    """Framing (no window is applied, self5:95) + rFFT + |X| and |X|^2/nfft.

    Returns (mag[F,K], pow[F,K], hamming_window, num_fft)   (self5:70-100).
    ``num_fft`` overrides the derived FFT length, for callers that pass nfft=512 to their MFCC
    routine at 8 kHz (inference_ctc_eri_05.py:253-257); the reference's winfft has no such argument.
    """
    audio = np.asarray(audio)
    size, step, nfft = frame_geometry(fs, frame_time, step_time)
    if num_fft is not None:
        nfft = int(num_fft)
    n_frames = frame_count(audio.shape[0], size, step)
    padded_len = (n_frames - 1) * step + size
    padded = np.zeros(padded_len, dtype=np.float64)  # self5:89-90 zero tail
    padded[: audio.shape[0]] = audio
    win = np.hamming(size)  # computed, returned, never applied (self5:78,95)
    nbin = nfft // 2 + 1
    mag = np.empty((n_frames, nbin), dtype=np.float64)
    pw = np.empty((n_frames, nbin), dtype=np.float64)
    if loop:
        for t in range(n_frames):
            chunk = padded[t * step : t * step + size]
            m = np.absolute(np.fft.rfft(chunk, nfft))  # self5:96
            mag[t] = m
            pw[t] = (1.0 / nfft) * np.square(m)  # self5:97
    else:
        idx = np.arange(n_frames)[:, None] * step + np.arange(size)[None, :]
        mag[:] = np.absolute(np.fft.rfft(padded[idx], nfft, axis=1))
        pw[:] = (1.0 / nfft) * np.square(mag)
    return mag, pw, win, nfft


def mel_bin_edges(fs=FS, numbins=NUM_BINS, lowfreq=LOW_FREQ, num_fft=512, highfreq=None):
    """FFT-bin positions of the numbins+2 mel peaks (self5:104-109).  The reference fixes the upper
    edge at fs/2 (:103); ``highfreq`` is the argument of the CTC scripts' MFCC routine
    (inference_ctc_eri_05.py:253-257), not part of the reference's melbanks."""
    highfreq = fs / 2 if highfreq is None else highfreq
    mel_lo = 2595 * np.log10(1 + lowfreq / 700.0)
    mel_hi = 2595 * np.log10(1 + highfreq / 700.0)
    mel_pts = np.linspace(mel_lo, mel_hi, numbins + 2)
    hz = 700 * (10 ** (mel_pts / 2595.0) - 1)
    return np.floor((num_fft + 1) * hz / fs)


def melbanks(fs=FS, numbins=NUM_BINS, lowfreq=LOW_FREQ, num_fft=512, highfreq=None):
    """Triangular mel filterbank [numbins, num_fft//2+1]  (self5:102-117).

    Rising edge (i-b_j)/(b_{j+1}-b_j) on [b_j, b_{j+1}); falling edge
    (b_{j+2}-i)/(b_{j+2}-b_{j+1}) on [b_{j+1}, b_{j+2}); no normalisation.
    """
    edges = mel_bin_edges(fs, numbins, lowfreq, num_fft, highfreq)
    bank = np.zeros((numbins, num_fft // 2 + 1))
    for j in range(numbins):
        lo, mid, hi = edges[j], edges[j + 1], edges[j + 2]
        for i in range(int(lo), int(mid)):
            bank[j, i] = (i - lo) / (mid - lo)
        for i in range(int(mid), int(hi)):
            bank[j, i] = (hi - i) / (hi - mid)
    return bank


def melspec(spec, melbank, loop=True):
    """log mel energies (UNfloored, self5:125-127: the np.where result is
    discarded) and floored log energy (self5:130-132).

    ``spec`` is the power list in self5 (:217) and the magnitude list in
    self15 (:234).  Returns (logmel[F,nmel], logE[F]).
    """
    spec = np.asarray(spec, dtype=np.float64)
    n_frames = spec.shape[0]
    nmel = melbank.shape[0]
    logmel = np.empty((n_frames, nmel))
    loge = np.empty(n_frames)
    with np.errstate(divide="ignore", invalid="ignore"):
        if loop:
            bank_t = melbank.T
            for t in range(n_frames):
                logmel[t] = np.log(np.dot(spec[t], bank_t))
                e = np.sum(spec[t], 0)
                if e == 0:
                    e = EPS64
                loge[t] = np.log(e)
        else:
            logmel[:] = np.log(spec @ melbank.T)
            e = spec.sum(axis=1)
            loge[:] = np.log(np.where(e == 0, EPS64, e))
    return logmel, loge


def dct_ortho(logmel, numcep=12, loop=True):
    """DCT-II, norm='ortho', first numcep coefficients (self5:136-142)."""
    logmel = np.asarray(logmel, dtype=np.float64)
    if loop:
        return np.stack([_scipy_dct(row, type=2, axis=0, norm="ortho")[:numcep] for row in logmel])
    return _scipy_dct(logmel, type=2, axis=1, norm="ortho")[:, :numcep]


def dct_matrix(nmel=NUM_BINS, numcep=13):
    """Closed form of the same transform: c = x @ W, W[nmel, numcep]."""
    n = np.arange(nmel)[:, None]
    k = np.arange(numcep)[None, :]
    w = np.cos(np.pi * k * (2 * n + 1) / (2 * nmel)) * np.sqrt(2.0 / nmel)
    w[:, 0] = np.sqrt(1.0 / nmel)
    return w


def lifter_weights(ncoeff, lift_factor=LIFT_FACTOR):
    """1 + (L/2) sin(pi n / L)   (self5:149-151); identity for L <= 0 (:153-155)."""
    if lift_factor > 0:
        n = np.arange(ncoeff)
        return 1 + (lift_factor / 2.0) * np.sin(np.pi * n / lift_factor)
    return np.ones(ncoeff)


def dctopt(cep, lift_factor=LIFT_FACTOR):
    cep = np.asarray(cep, dtype=np.float64)
    if lift_factor > 0:
        return lifter_weights(cep.shape[1], lift_factor) * cep
    return cep.copy()


def staticfeatures(loge, lifted):
    """Coefficient 0 is REPLACED by log energy after liftering (self5:158-167)."""
    out = np.array(lifted, dtype=np.float64, copy=True)
    out[:, 0] = loge
    return out


def delta(features, pad_fac=2, loop=True):
    """Regression delta with edge replication (self5:169-179)."""
    features = np.asarray(features, dtype=np.float64)
    n_frames = features.shape[0]
    denom = 2 * sum(i * i for i in range(1, pad_fac + 1))
    padded = np.pad(features, ((pad_fac, pad_fac), (0, 0)), mode="edge")
    taps = np.arange(-pad_fac, pad_fac + 1)
    out = np.zeros_like(features)
    if loop:
        for t in range(n_frames):
            out[t] = np.dot(taps, padded[t : t + 2 * pad_fac + 1]) / denom
    else:
        acc = np.zeros_like(features)
        for j, k in enumerate(taps):
            acc += k * padded[j : j + n_frames]
        out = acc / denom
    return out


def spectrograms(cep, d, dd):
    """[c | delta | delta-delta] column concat (self5:181-186)."""
    return np.concatenate([cep, d, dd], axis=1)


# --------------------------------------------------------------------------
# whole pipelines (self5:211-229, self15:228-246)
# --------------------------------------------------------------------------
def mfcc39(audio, variant="self5", fs=FS, frame_time=FRAME_TIME, step_time=STEP_TIME,
           num_bins=NUM_BINS, low_freq=LOW_FREQ, lift_factor=LIFT_FACTOR, loop=True,
           return_stages=False, num_fft=None):
    cfg = VARIANTS[variant]
    audio = np.asarray(audio)
    sig = preemphasis(audio, cfg["preemph"]) if cfg["preemph"] is not None else audio
    mag, pw, _win, nfft = winfft(sig, fs, frame_time, step_time, loop=loop, num_fft=num_fft)
    bank = melbanks(fs, num_bins, low_freq, nfft)
    spec = pw if cfg["spectrum"] == "power" else mag
    logmel, loge = melspec(spec, bank, loop=loop)
    with np.errstate(invalid="ignore"):
        cep = dct_ortho(logmel, 13, loop=loop)
        lifted = dctopt(cep, lift_factor)
        static = staticfeatures(loge, lifted)
        d = delta(static, loop=loop)
        dd = delta(d, loop=loop)
    out = spectrograms(static, d, dd)
    if return_stages:
        return dict(signal=np.asarray(sig, dtype=np.float64), mag=mag, pow=pw, melbank=bank,
                    logmel=logmel, loge=loge, cep=cep, lifted=lifted, static=static,
                    delta=d, delta2=dd, out=out)
    return out


def spectrum257(audio, variant="self5", kind=None, log=False, loop=False, fs=FS, frame_time=FRAME_TIME,
                step_time=STEP_TIME, num_fft=None):
    """The 257-bin variant: winfft's per-frame spectrum stacked to [F,257]
    (self5:96-99); optional natural log of it."""
    cfg = VARIANTS[variant]
    audio = np.asarray(audio)
    sig = preemphasis(audio, cfg["preemph"]) if cfg["preemph"] is not None else audio
    mag, pw, _w, _n = winfft(sig, fs, frame_time, step_time, loop=loop, num_fft=num_fft)
    spec = pw if (kind or cfg["spectrum"]) == "power" else mag
    if log:
        with np.errstate(divide="ignore"):
            spec = np.log(spec)
    return spec


# --------------------------------------------------------------------------
# context stacking + .pat container (MLP_5hidlayer_40.py:61-105)
# --------------------------------------------------------------------------
def stack_context(feats, framenum):
    """[F,C] -> [F, framenum*C], frame-major, edges replicated.

    Equivalent to padUtterance (MLP_5hidlayer_40.py:70-87: (N-1)/2 copies of
    the first/last frame on each side) followed by slidingWin (:89-105:
    reshape of the [C,N] slice in Fortran order = frame-major) on the
    transposed [C,F] data the reference carries.
    out[t, f*C + c] = feats[clamp(t + f - (N-1)//2), c].
    """
    feats = np.asarray(feats)
    n_frames, _c = feats.shape
    half = int((framenum - 1) / 2)
    rows = []
    for t in range(n_frames):
        idx = np.clip(np.arange(t - half, t - half + framenum), 0, n_frames - 1)
        rows.append(feats[idx].reshape(-1))
    return np.stack(rows)


def stack_context_tiles(feats, framenum, order=None):
    """The CNN scripts' form of the same windows (CNN_DenseNet40_6.py:86-103): each window is
    kept as the [C, N] slice paddeddata[:, pos:pos+framenum] (:98), visited in ``order``
    (createRandomvec(randomint, F) when randomint != -11, :91-97).  Returns [len(order), C, N]:
    out[i, c, f] = feats[clamp(order[i] + f - (N-1)//2), c]."""
    feats = np.asarray(feats)
    n_frames, _c = feats.shape
    half = int((framenum - 1) / 2)
    order = np.arange(n_frames) if order is None else np.asarray(order)
    tiles = []
    for t in order:
        idx = np.clip(np.arange(t - half, t - half + framenum), 0, n_frames - 1)
        tiles.append(feats[idx].T)
    return np.stack(tiles) if len(tiles) else np.zeros((0, feats.shape[1], framenum), feats.dtype)


def random_order(randomint, n):
    """createRandomvec (MLP_5hidlayer_40.py:107-112, CNN_DenseNet40_6.py:105-110)."""
    np.random.seed(randomint)
    vec = np.arange(n)
    np.random.shuffle(vec)
    return vec


def read_pat(path_or_bytes, coeffs):
    """HTK-style container: 12-byte header {int32 nFrames, int32 sampPeriod,
    int16 sampSize, int16 parmKind} + float32[nFrames, coeffs] row-major
    (MLP_5hidlayer_40.py:61-68 reads nFrames from the first int32 and skips
    12 bytes).  Returns (feats[F,C] float32, header tuple)."""
    if isinstance(path_or_bytes, (bytes, bytearray)):
        raw = bytes(path_or_bytes)
    else:
        with open(path_or_bytes, "rb") as fh:
            raw = fh.read()
    n_frames, period, samp_size, kind = struct.unpack("<iihh", raw[:12])
    data = np.frombuffer(raw, dtype="<f4", offset=12, count=n_frames * coeffs)
    return data.reshape(n_frames, coeffs).copy(), (n_frames, period, samp_size, kind)


def write_pat(feats, period=100000, kind=838):
    feats = np.ascontiguousarray(feats, dtype="<f4")
    n_frames, coeffs = feats.shape
    return struct.pack("<iihh", n_frames, period, coeffs * 4, kind) + feats.tobytes()


# --------------------------------------------------------------------------
# CMVN (generalisation of inference_ctc_eri_05.py:291 / anal_recog_nn_liste.m:180)
# --------------------------------------------------------------------------
def cmvn_sums(feats):
    """(n, sum[D], sumsq[D]) in float64."""
    feats = np.asarray(feats, dtype=np.float64)
    return feats.shape[0], feats.sum(axis=0), np.square(feats).sum(axis=0)


def cmvn_constants(n, s, ss, ddof=0):
    """mean, 1/std with the (x + mm) * vv form of anal_recog_nn_liste.m:180."""
    mean = s / n
    var = (ss - n * mean * mean) / (n - ddof)
    return -mean, 1.0 / np.sqrt(var)


def cmvn_apply(feats, mm, vv):
    return (np.asarray(feats, dtype=np.float64) + mm) * vv


def utterance_scalar_norm(feats):
    """(x - mean(x)) / std(x) over ALL elements (inference_ctc_eri_05.py:291)."""
    feats = np.asarray(feats, dtype=np.float64)
    return (feats - np.mean(feats)) / np.std(feats)


def pad_sequences(sequences, maxlen=None, padding="post", truncating="post", value=0.0):
    """pad_sequences of CTC/utils.py:71-128 for lists of [F_i, D] arrays:
    returns (x[B, maxlen, D] float32, lengths int64[B])."""
    lengths = np.asarray([len(s) for s in sequences], dtype=np.int64)
    if maxlen is None:
        maxlen = int(lengths.max())
    shape = next((np.asarray(s).shape[1:] for s in sequences if len(s) > 0), ())
    x = (np.ones((len(sequences), maxlen) + tuple(shape)) * value).astype(np.float32)
    for i, s in enumerate(sequences):
        if len(s) == 0:
            continue
        trunc = np.asarray(s[-maxlen:] if truncating == "pre" else s[:maxlen], dtype=np.float32)
        if padding == "post":
            x[i, :len(trunc)] = trunc
        else:
            x[i, -len(trunc):] = trunc
    return x, lengths


def spec_anal(sig, bpre, framelength=400, frameshift=160, nfft=512, use_preemp=True):
    """spec_anal of anal_recog_nn_liste.m:497-532 for fs > 8000: filter(bpre, 1, sig) with its
    final state appended and the group delay dropped (:511-516), hamming(framelength) frames,
    fix((L - noverlap)/frameshift) of them (:503), log(max(|fft|, 1)) (:529)."""
    from scipy.signal import lfilter
    sig = np.asarray(sig, dtype=np.float64)
    if use_preemp:
        b = np.asarray(bpre, dtype=np.float64)
        y, z = lfilter(b, [1.0], sig, zi=np.zeros(len(b) - 1))
        gd = (len(b) - 1) // 2
        sig = np.concatenate([y[gd:], z[:gd]])
    nframes = max(0, int((len(sig) - (framelength - frameshift)) / frameshift))
    win = np.hamming(framelength)
    spec = np.zeros((nframes, nfft // 2 + 1))
    for t in range(nframes):
        frame = win * sig[t * frameshift: t * frameshift + framelength]
        spec[t] = np.log(np.maximum(np.abs(np.fft.fft(frame, nfft)[: nfft // 2 + 1]), 1.0))
    return spec
