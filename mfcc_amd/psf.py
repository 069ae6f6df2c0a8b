"""Entry points with the call signature of the ``python_speech_features`` package, which the
reference's CTC / word-label scripts call (inference_ctc_eri_05.py:253-257,
CTC/ctc_test_1.py:51, CNN_wordlabel_2.py:149):

    mfcc_feat = mfcc(audio, samplerate=fs, winlen=0.025, winstep=0.01, numcep=13, nfilt=26,
                     nfft=512, lowfreq=0, highfreq=None, preemph=0.97, ceplifter=22,
                     appendEnergy=True)
    d = delta(mfcc_feat, 2)

The package itself is not available to check against, so parity with it is UNPINNED; what is
checked is the documented algorithm, which is the numpy routine of mfcc-self5.py with two
differences: mel energies that are exactly zero are floored to eps before the log (the numpy
routine discards that floor, mfcc-self5.py:126-127), and ``appendEnergy=False`` keeps the DCT's
own c0.  Frame sizes are int(winlen * samplerate) as the reference's winfft takes them
(mfcc-self5.py:75-76; the package rounds half up, which differs only when winlen * samplerate has a
fractional part of .5 or more).  25 ms / 10 ms frames at 16 kHz or 8 kHz (RATE = 8000 in
inference_ctc_eri_05.py:96) with nfft 512 run the specialised kernel; every other framing, FFT
length (a power of two), highfreq, or appendEnergy=False runs the general one.
"""
from __future__ import annotations

import numpy as np
import torch

from . import frontend as F


def mfcc(signal, samplerate=16000, winlen=0.025, winstep=0.01, numcep=13, nfilt=26, nfft=512, lowfreq=0,
         highfreq=None, preemph=0.97, ceplifter=22, appendEnergy=True, winfunc=None):
    """[L] samples -> [F, numcep] float32 cepstra (c0 replaced by ln of the frame energy when
    ``appendEnergy``)."""
    size = int(samplerate * (winlen * 1000 * (1.0 / 1000)))
    window = None
    if winfunc is not None:
        w = np.asarray(winfunc(size), dtype=np.float64)
        if w.shape != (size,):
            raise ValueError("winfunc(frame_length) must return frame_length weights")
        if np.array_equal(w, np.hamming(size)):
            window = "hamming"
        elif not np.array_equal(w, np.ones(size)):
            raise ValueError("winfunc: the rectangular window (package default) and numpy.hamming are implemented")
    return F.mfcc39(signal, fs=samplerate, frame_time=winlen * 1000, step_time=winstep * 1000, num_bins=nfilt,
                    low_freq=lowfreq, high_freq=highfreq, numcep=numcep, lift_factor=ceplifter,
                    preemph=preemph if preemph else None, spectrum="power", floor_mel=True, kind="static",
                    num_fft=nfft, window=window, keep_c0=not appendEnergy)


def delta(feat, N):
    """Regression over +-N frames with edge replication (the delta of mfcc-self5.py:169-179 with
    pad_fac = N)."""
    if N < 1:
        raise ValueError("N must be an integer >= 1")
    return F.delta(feat, int(N))


def get_wav_mfcc(audio, fs=16000):
    """get_wav_mfcc of inference_ctc_eri_05.py:245-263 for an already loaded signal:
    [mfcc | delta | delta-delta] = [F, 39] in one fused launch."""
    return F.mfcc39(audio, fs=fs, floor_mel=True, num_fft=512)
