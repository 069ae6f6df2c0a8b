"""The 'plain DFT analysis' front-end of the reference's Matlab recogniser, spec_anal
(anal_recog_nn_liste.m:497-532), which produced the 257-coefficient training data of the MLP/CNN
scripts: 15-tap FIR pre-emphasis with its group delay removed (:511-516), Hamming(400) frames every
160 samples, complete frames only (:503), 512-point FFT, log(max(|X|, 1)) (:529), followed by the
corpus normalisation (x + mm) .* vv (:180).

No Matlab is available to check against: parity with the reference is UNPINNED; the tests hold the
result to a numpy/scipy restatement of the lines cited above.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _capi as capi
from . import frontend as F

# initdft for fs = 16000 (anal_recog_nn_liste.m:52-57)
initdft = dict(
    fs=16000, framelength=400, frameshift=160, noverlap=240, nfft=512, use_preemp=True,
    bpre=[0.0114, -0.0037, -0.0141, 0.0009, 0.0447, -0.0114, -0.1880, 1.4774, -0.6140,
          -0.4631, -0.2080, -0.0028, 0.0365, 0.0035, -0.01951],
)


def fir_centered(audio, lengths, taps, delay, plan=None):
    """out[n] = sum_k taps[k] x[n + delay - k] per utterance, zero outside it: filter() plus the
    group-delay shift of anal_recog_nn_liste.m:511-516.  audio: packed device tensor (int16 or
    float32); returns float32."""
    plan = plan or F.get_plan(audio.device)
    lay = F.Layout(plan, lengths)
    taps = np.ascontiguousarray(np.asarray(taps, dtype=np.float32))
    out = torch.empty(audio.numel(), dtype=torch.float32, device=audio.device)
    code = capi.AUDIO_I16 if audio.dtype == torch.int16 else capi.AUDIO_F32
    with F._on(plan.device, audio):
        capi.check(capi.load().mfcc_fir(lay._h, F._ptr(audio), code, taps.ctypes.data_as(C.POINTER(C.c_float)), len(taps),
                                        int(delay), F._ptr(out), F._stream_ptr()))
    return out


def spec_anal(sig, init=None, lengths=None, mm=None, vv=None):
    """[L] samples (or a packed batch with ``lengths``) -> float32 [F, nfft/2+1] log-magnitude
    spectrum rows, F = fix((L - noverlap) / frameshift) per utterance.  ``mm`` / ``vv`` apply the
    corpus normalisation (x + mm) .* vv in place (anal_recog_nn_liste.m:180)."""
    init = dict(initdft, **(init or {}))
    if (init["framelength"], init["frameshift"], init["nfft"], init["fs"]) != (400, 160, 512, 16000):
        raise ValueError("spec_anal is built for initdft at fs = 16000: 400/160-sample frames, 512-point FFT")
    t, _code, was_numpy = F._audio_tensor(sig)
    flat = t.reshape(-1)
    if lengths is None:
        lengths = [t.shape[-1]] * (t.shape[0] if t.dim() == 2 else 1)
    plan = F.get_plan(flat.device, preemph=None, spectrum="mag", window="hamming", frame_rule="complete", log_floor=1.0)
    if init["use_preemp"]:
        taps = init["bpre"]
        flat = fir_centered(flat, lengths, taps, (len(taps) - 1) // 2, plan)   # group delay shift only above 8 kHz (:513)
    fe = F.MFCC.__new__(F.MFCC)
    fe.plan, fe._layouts = plan, F._layout_cache.setdefault(id(plan), {})
    spec = fe.forward(flat, fe.layout(lengths), kind="logspec")
    if mm is not None:
        F.cmvn_apply(spec, F._to_dev(np.asarray(mm), torch.float32, spec.device), F._to_dev(np.asarray(vv), torch.float32, spec.device))
    return F._ret(spec, was_numpy)
