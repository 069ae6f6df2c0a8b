"""mfcc_amd -- MI355X (gfx950) implementation of the Robomate/mfcc MFCC front-end.

``from mfcc_amd import *`` provides the reference's function names (mfcc-self5.py /
mfcc-self15.py, MLP_5hidlayer_40.py) backed by HIP kernels, plus the fused batch API.
"""
from .frontend import (  # noqa: F401
    MFCC, Layout, Plan, StreamingMFCC, VARIANTS, cmvn_accumulate, cmvn_allreduce, cmvn_apply, cmvn_constants, cmvn_release, coeffs, dctopt, delta,
    delta_factor, duration, frame_time, fs, get_plan, lift_factor, loadPatData, loadrawdata, low_freq, melbanks,
    melspec, mfcc, mfcc39, mfcc39_stack, num_bins, num_fft, padUtterance, preemphasis, shard_utterances, slidingWin, spectrograms,
    spectrum257, spectrum_stack, stack_context, staticfeatures, step_time, winfft, writePatData)

__all__ = [
    "MFCC", "Layout", "Plan", "StreamingMFCC", "VARIANTS", "mfcc39", "mfcc39_stack", "spectrum257", "spectrum_stack", "stack_context", "cmvn_accumulate", "cmvn_release",
    "cmvn_allreduce", "cmvn_constants", "cmvn_apply", "shard_utterances", "get_plan",
    # reference names
    "loadrawdata", "preemphasis", "winfft", "melbanks", "melspec", "mfcc", "dctopt", "staticfeatures", "delta",
    "spectrograms", "loadPatData", "writePatData", "padUtterance", "slidingWin",
    "frame_time", "step_time", "num_bins", "low_freq", "num_fft", "lift_factor", "delta_factor", "fs", "duration",
]
from . import cnn, corpus, datasets, psf, sequences  # noqa: E402,F401  (counterparts of the reference's data plumbing / call sites)
from .corpus import corpus_mfcc_cmvn  # noqa: E402,F401
from .sequences import ctc_feature_batch, pad_sequences, utterance_scalar_norm  # noqa: E402,F401
