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
own c0.  Supported geometry is what the gfx950 kernels are built for: 25 ms / 10 ms frames at
16 kHz or at 8 kHz (RATE = 8000 in inference_ctc_eri_05.py:96), nfft 512, highfreq = samplerate / 2.
"""
from __future__ import annotations

import numpy as np
import torch

from . import frontend as F


def mfcc(signal, samplerate=16000, winlen=0.025, winstep=0.01, numcep=13, nfilt=26, nfft=512, lowfreq=0,
         highfreq=None, preemph=0.97, ceplifter=22, appendEnergy=True, winfunc=None):
    """[L] samples -> [F, numcep] float32 cepstra (c0 replaced by ln of the frame energy when
    ``appendEnergy``)."""
    if highfreq is not None and float(highfreq) != samplerate / 2:
        raise ValueError("highfreq must be None or samplerate/2 (the filterbank ends at the Nyquist frequency)")
    if nfft != 512:
        raise ValueError("nfft must be 512")
    if winfunc is not None:
        raise ValueError("winfunc: only the package default (rectangular window) is implemented")
    if not appendEnergy:
        raise ValueError("appendEnergy=False is not implemented (c0 is always ln of the frame energy)")
    return F.mfcc39(signal, fs=samplerate, frame_time=winlen * 1000, step_time=winstep * 1000, num_bins=nfilt,
                    low_freq=lowfreq, numcep=numcep, lift_factor=ceplifter, preemph=preemph if preemph else None,
                    spectrum="power", floor_mel=True, kind="static", num_fft=nfft)


def delta(feat, N):
    """Regression over +-N frames with edge replication; N = 2 is implemented (every call site of
    the reference uses 2)."""
    if N != 2:
        raise ValueError("delta: N=2 is implemented")
    return F.delta(feat, N)


def get_wav_mfcc(audio, fs=16000):
    """get_wav_mfcc of inference_ctc_eri_05.py:245-263 for an already loaded signal:
    [mfcc | delta | delta-delta] = [F, 39] in one fused launch."""
    return F.mfcc39(audio, fs=fs, floor_mel=True, num_fft=512)
