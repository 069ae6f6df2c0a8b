"""Batch shaping for the recurrent / CTC consumers of the features, on the device.

The reference's scripts do this per utterance in host loops: zero-padded
``[batch, time_steps, coeffs]`` arrays plus sequence lengths (BiLSTM_7.py:82-124,
RNN_CTC_17.py:82-155, ``pad_sequences`` in CTC/utils.py:71-128) and the per-utterance scalar
normalisation ``(x - mean(x)) / std(x)`` (inference_ctc_eri_05.py:291, CTC/ctc_test_1.py:54).
Here the ragged ``[sum F, D]`` output of the front-end goes straight to those shapes.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _capi as capi
from .frontend import MFCC, _on, _ptr, _stream_ptr, frames_layout


def _frames_layout(feats, frame_counts):
    frame_counts = np.asarray(frame_counts, dtype=np.int64).reshape(-1)
    if int(frame_counts.sum()) != feats.shape[0]:
        raise ValueError("frame_counts do not add up to the number of rows")
    return frames_layout(feats.device, frame_counts), frame_counts


def utterance_scalar_norm(feats, frame_counts):
    """In place: every utterance's rows become (x - mean) / std, mean and population std over
    all its frames and coefficients (inference_ctc_eri_05.py:291)."""
    if not feats.is_cuda or feats.dtype != torch.float32 or feats.dim() != 2 or feats.stride(1) != 1:
        raise ValueError("feats must be a float32 device tensor [rows, D] with unit column stride")
    lay, _ = _frames_layout(feats, frame_counts)
    with _on(feats.device):
        capi.check(capi.load().mfcc_utt_scalar_norm(lay._h, _ptr(feats), feats.stride(0), feats.shape[1], _stream_ptr()))
    return feats


def pad_sequences(feats, frame_counts, maxlen=None, padding="post", truncating="post", value=0.0, out=None):
    """Ragged rows -> (x [B, maxlen, D] float32, lengths int64[B]) with the semantics of
    pad_sequences (CTC/utils.py:71-128): maxlen defaults to the longest sequence; longer ones
    are truncated at the end ('post') or the beginning ('pre'); shorter ones are padded with
    ``value`` after ('post') or before ('pre').  ``lengths`` are the original lengths."""
    if padding not in ("pre", "post"):
        raise ValueError('Padding type "%s" not understood' % padding)
    if truncating not in ("pre", "post"):
        raise ValueError('Truncating type "%s" not understood' % truncating)
    if not feats.is_cuda or feats.dtype != torch.float32 or feats.dim() != 2 or feats.stride(1) != 1:
        raise ValueError("feats must be a float32 device tensor [rows, D] with unit column stride")
    lay, frame_counts = _frames_layout(feats, frame_counts)
    if maxlen is None:
        maxlen = int(frame_counts.max()) if len(frame_counts) else 0
    b, d = len(frame_counts), feats.shape[1]
    if out is None:
        out = torch.empty((b, maxlen, d), dtype=torch.float32, device=feats.device)
    elif tuple(out.shape) != (b, maxlen, d) or out.dtype != torch.float32 or not out.is_contiguous():
        raise ValueError("out must be a contiguous float32 [B, maxlen, D] device tensor")
    if b and maxlen:
        with _on(feats.device, out):
            capi.check(capi.load().mfcc_pad_batch(lay._h, _ptr(feats), feats.stride(0), d, int(maxlen), float(value),
                                                  int(padding == "pre"), int(truncating == "pre"), _ptr(out), _stream_ptr()))
    return out, frame_counts.copy()


def ctc_feature_batch(audios, variant="self5", time_steps=None, normalize=True, device=None, **overrides):
    """Feature side of create_minibatch_ctc (inference_ctc_eri_05.py:283-300): a list of
    utterances (int16 / float arrays of any lengths) -> (features [B, T, 39], seq_length
    int64[B]) with per-utterance scalar normalisation and zero padding: three launches on a
    repeated batch shape (the offset tables of a new shape are uploaded once and cached)."""
    fe = MFCC(variant, device=device, **overrides)
    lengths = [int(np.asarray(a).shape[-1]) if not isinstance(a, torch.Tensor) else int(a.shape[-1]) for a in audios]
    parts = [a if isinstance(a, torch.Tensor) else torch.as_tensor(np.ascontiguousarray(a)) for a in audios]
    dtype = torch.int16 if all(p.dtype == torch.int16 for p in parts) else torch.float32
    flat = torch.cat([p.reshape(-1).to(fe.plan.device).to(dtype) for p in parts])
    lay = fe.layout(lengths)
    feats = fe.forward(flat, lay)
    counts = np.diff(lay.frame_offsets)
    if normalize:
        utterance_scalar_norm(feats, counts)
    return pad_sequences(feats, counts, maxlen=time_steps)
