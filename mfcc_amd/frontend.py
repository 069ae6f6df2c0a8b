"""Host side of the MI355X MFCC front-end.

Two layers over the C ABI (include/mfcc_amd.h, bound in _capi.py):

* the fused fast path -- :class:`MFCC`, :func:`mfcc39`, :func:`spectrum257`,
  :func:`stack_context`, CMVN helpers -- one kernel launch per batch of utterances;
* the reference's own function names and signatures (``preemphasis``, ``winfft``,
  ``melbanks``, ``melspec``, ``mfcc``, ``dctopt``, ``staticfeatures``, ``delta``,
  ``spectrograms``, ``loadrawdata``; ``loadPatData``, ``padUtterance``, ``slidingWin``)
  so that the call chain of ``main()`` in mfcc-self5.py:211-229 /
  mfcc-self15.py:228-246 and ``Preprocess`` in MLP_5hidlayer_40.py:144-161 runs
  unchanged with ``from mfcc_amd import *``.

PyTorch is used only for device buffers and the current HIP stream.  Every
numeric result comes from the HIP kernels; nothing here computes features on
the CPU.  numpy in -> numpy out (float32), torch in -> torch (device) out.
"""
from __future__ import annotations

import ctypes as C
import struct

import numpy as np
import torch

from . import _capi as capi

# the reference's module-level globals (mfcc-self5.py:21-29)
frame_time = 25
step_time = 10
num_bins = 26
low_freq = 0
num_fft = 512
lift_factor = 22
delta_factor = 2
fs = 16000
duration = 5

VARIANTS = {
    "self5": dict(preemph=0.97, spectrum="power"),   # mfcc-self5.py:211-217
    "self15": dict(preemph=None, spectrum="mag"),    # mfcc-self15.py:228-234
}
_OUT_KINDS = {"mfcc39": capi.OUT_MFCC39, "static": capi.OUT_STATIC, "logmel": capi.OUT_LOGMEL,
              "spec": capi.OUT_SPEC, "logspec": capi.OUT_LOGSPEC}


# --------------------------------------------------------------------------
# small helpers
# --------------------------------------------------------------------------
def _device(device=None):
    if not torch.cuda.is_available():
        raise RuntimeError("mfcc_amd needs a HIP device (torch.cuda.is_available() is False); there is no CPU path")
    if device is None:
        return torch.device("cuda", torch.cuda.current_device())
    dev = torch.device(device)
    if dev.type != "cuda":
        raise ValueError(f"mfcc_amd runs on HIP devices only, got {dev}")
    if dev.index is None:   # 'cuda' = the current device
        dev = torch.device("cuda", torch.cuda.current_device())
    return dev


class _NoSwitch:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


_NO_SWITCH = _NoSwitch()


def _on(dev, *tensors):
    """Context for one launch on ``dev``: every tensor handed to the library must live there,
    and the device is made current so that the stream passed down is that device's current
    stream (a kernel enqueued on another device's stream would dereference foreign memory)."""
    for t in tensors:
        if t is not None and t.device != dev:
            raise ValueError(f"tensor on {t.device}, but the plan / layout / other operands live on {dev}")
    if torch.cuda.current_device() == dev.index:
        return _NO_SWITCH   # already current: entering torch.cuda.device() costs several microseconds per launch
    return torch.cuda.device(dev)


def _stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return C.c_void_p(t.data_ptr())


def _as_rows(x):
    """list of per-frame arrays / 2-D array / tensor -> (tensor-or-array 2-D, was_numpy)."""
    if isinstance(x, torch.Tensor):
        return x, False
    if isinstance(x, (list, tuple)) and len(x) and isinstance(x[0], torch.Tensor):
        return torch.stack(list(x)), False
    return np.asarray(x), True


def _to_dev(x, dtype, device=None):
    if isinstance(x, torch.Tensor):
        t = x if x.is_cuda else x.to(_device(device))
        return t.to(dtype).contiguous()
    return torch.as_tensor(np.ascontiguousarray(x)).to(_device(device)).to(dtype).contiguous()


def _ret(t, was_numpy):
    return t.cpu().numpy() if was_numpy else t


def _audio_tensor(audio, device=None):
    """-> (contiguous device tensor int16|float32, dtype code, was_numpy)."""
    was_numpy = not isinstance(audio, torch.Tensor)
    if was_numpy:
        a = np.asarray(audio)
        if a.dtype == np.int16:
            t = torch.from_numpy(np.ascontiguousarray(a))
        else:
            t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
        t = t.to(_device(device))
    else:
        t = audio if audio.is_cuda else audio.to(_device(device))
        if t.dtype != torch.int16:
            t = t.to(torch.float32)
        t = t.contiguous()
    code = capi.AUDIO_I16 if t.dtype == torch.int16 else capi.AUDIO_F32
    return t, code, was_numpy


# --------------------------------------------------------------------------
# plan / layout objects
# --------------------------------------------------------------------------
class Plan:
    """Immutable constant tables for one configuration on one device."""

    def __init__(self, fs=16000, frame_time=25, step_time=10, num_bins=26, low_freq=0, numcep=13,
                 lift_factor=22, pad_fac=2, preemph=0.97, spectrum="power", window=None, floor_mel=False,
                 num_fft=None, frame_rule="pad", log_floor=0.0, high_freq=None, kernel="auto", keep_c0=False, device=None):
        self.device = _device(device)
        lib = capi.load()
        cfg = capi.MfccConfig()
        lib.mfcc_config_default(C.byref(cfg))
        cfg.fs, cfg.frame_time, cfg.step_time = float(fs), float(frame_time), float(step_time)
        cfg.num_bins, cfg.low_freq, cfg.numcep = int(num_bins), float(low_freq), int(numcep)
        cfg.lift_factor, cfg.pad_fac = float(lift_factor), int(pad_fac)
        cfg.use_preemph = 0 if preemph is None else 1
        cfg.preemph = 0.0 if preemph is None else float(preemph)
        if spectrum not in ("power", "mag"):
            raise ValueError("spectrum must be 'power' or 'mag'")
        cfg.spectrum = capi.SPECTRUM_POWER if spectrum == "power" else capi.SPECTRUM_MAG
        if window not in (None, "hamming"):
            raise ValueError("window must be None or 'hamming'")
        cfg.window = capi.WINDOW_NONE if window is None else capi.WINDOW_HAMMING
        cfg.floor_mel = 1 if floor_mel else 0
        cfg.num_fft = 0 if num_fft is None else int(num_fft)   # None: next power of two >= frame size (winfft)
        if frame_rule not in ("pad", "complete"):
            raise ValueError("frame_rule must be 'pad' (winfft) or 'complete' (spec_anal)")
        cfg.frame_rule = 0 if frame_rule == "pad" else 1
        cfg.log_floor = float(log_floor)
        cfg.high_freq = 0.0 if high_freq is None else float(high_freq)   # None: fs/2 as melbanks (mfcc-self5.py:103)
        if kernel not in ("auto", "general"):
            raise ValueError("kernel must be 'auto' or 'general'")
        cfg.kernel = capi.KERNEL_AUTO if kernel == "auto" else capi.KERNEL_GENERAL
        cfg.keep_c0 = 1 if keep_c0 else 0
        self.cfg = cfg
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            capi.check(lib.mfcc_plan_create(C.byref(cfg), self.device.index, C.byref(h)))
        self._h = h
        v = [C.c_int32() for _ in range(5)]
        capi.check(lib.mfcc_plan_info(h, *[C.byref(x) for x in v]))
        self.frame_size, self.frame_step, self.num_fft, self.num_bins, self.numcep = [x.value for x in v]
        # True: the general kernel runs (a framing, pad_fac or filterbank outside the specialised one)
        self.general = capi.check(lib.mfcc_plan_kernel(h)) == capi.KERNEL_GENERAL

    def out_width(self, kind):
        return capi.check(capi.load().mfcc_out_width(self._h, _OUT_KINDS[kind]))

    def frames_for_length(self, n):
        if self.cfg.frame_rule == 1:   # complete frames only (anal_recog_nn_liste.m:503)
            return max(0, (int(n) - (self.frame_size - self.frame_step)) // self.frame_step)
        return int(capi.check(capi.load().mfcc_frames_for_length(int(n), self.frame_size, self.frame_step)))

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                capi.load().mfcc_plan_destroy(h)
            except Exception:
                pass


class Layout:
    """Geometry of a ragged batch: utterance lengths -> sample/frame offsets and the
    tile list of the fused kernel (device resident)."""

    def __init__(self, plan, lengths):
        lengths = np.ascontiguousarray(np.asarray(lengths, dtype=np.int64).reshape(-1))
        self.plan = plan
        self.device = plan.device
        self.lengths = lengths
        lib = capi.load()
        h = C.c_void_p()
        with torch.cuda.device(plan.device):
            capi.check(lib.mfcc_layout_create(plan._h, lengths.ctypes.data_as(C.POINTER(C.c_int64)), len(lengths),
                                              C.byref(h)))
        self._h = h
        n, ts, tf = C.c_int64(), C.c_int64(), C.c_int64()
        capi.check(lib.mfcc_layout_info(h, C.byref(n), C.byref(ts), C.byref(tf)))
        self.n_utts, self.total_samples, self.total_frames = n.value, ts.value, tf.value
        so = np.zeros(len(lengths) + 1, dtype=np.int64)
        fo = np.zeros(len(lengths) + 1, dtype=np.int64)
        capi.check(lib.mfcc_layout_offsets_host(h, so.ctypes.data_as(C.POINTER(C.c_int64)),
                                                fo.ctypes.data_as(C.POINTER(C.c_int64))))
        self.sample_offsets, self.frame_offsets = so, fo

    @classmethod
    def from_frame_counts(cls, plan_or_device, frame_counts):
        """Layout whose utterances have the given FRAME counts (>= 0), for feature rows that
        did not come from audio (e.g. .pat files).  Holds frame offsets only."""
        dev = plan_or_device.device if isinstance(plan_or_device, Plan) else _device(plan_or_device)
        fc = np.ascontiguousarray(np.asarray(frame_counts, dtype=np.int64).reshape(-1))
        lay = cls.__new__(cls)
        lay.plan, lay.device, lay.lengths = None, dev, None
        lib = capi.load()
        h = C.c_void_p()
        with torch.cuda.device(dev):
            capi.check(lib.mfcc_layout_create_frames(dev.index, fc.ctypes.data_as(C.POINTER(C.c_int64)), len(fc), C.byref(h)))
        lay._h = h
        lay.n_utts, lay.total_samples, lay.total_frames = len(fc), 0, int(fc.sum())
        lay.sample_offsets = np.zeros(len(fc) + 1, dtype=np.int64)
        lay.frame_offsets = np.concatenate([[0], np.cumsum(fc)]).astype(np.int64)
        return lay

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                capi.load().mfcc_layout_destroy(h)
            except Exception:
                pass


_plan_cache = {}


def get_plan(device=None, **cfg):
    dev = _device(device)
    key = (dev.index,) + tuple(sorted(cfg.items()))
    p = _plan_cache.get(key)
    if p is None:
        p = _plan_cache[key] = Plan(device=dev, **cfg)
    return p


# --------------------------------------------------------------------------
# fused fast path
# --------------------------------------------------------------------------
class MFCC:
    """Batched MFCC extractor.  ``variant`` selects the reference parameterisation:
    'self5' (pre-emphasis 0.97, power spectrum; mfcc-self5.py:211-217) or 'self15'
    (no pre-emphasis, magnitude spectrum; mfcc-self15.py:230,234)."""

    def __init__(self, variant="self5", device=None, **overrides):
        cfg = dict(VARIANTS[variant])
        cfg.update(overrides)
        self.plan = get_plan(device, **cfg)
        self._layouts = {}

    def layout(self, lengths):
        lengths = np.asarray(lengths, dtype=np.int64).reshape(-1)
        key = lengths.tobytes()
        lay = self._layouts.get(key)
        if lay is None:
            if len(self._layouts) > 64:
                self._layouts.clear()
            lay = self._layouts[key] = Layout(self.plan, lengths)
        return lay

    def forward(self, audio, layout, out=None, kind="mfcc39"):
        """audio: contiguous device tensor (int16 or float32) holding layout.total_samples
        samples; out: optional float32 device tensor [layout.total_frames, width]."""
        if not audio.is_cuda or not audio.is_contiguous():
            raise ValueError("audio must be a contiguous device tensor")
        if audio.dtype not in (torch.int16, torch.float32):
            raise ValueError("audio must be int16 or float32")
        if audio.numel() != layout.total_samples:
            raise ValueError(f"audio has {audio.numel()} samples, layout expects {layout.total_samples}")
        width = self.plan.out_width(kind)
        if kind == "mfcc39" and not capi.check(capi.load().mfcc_plan_supports(self.plan._h, _OUT_KINDS[kind])):
            return self._mfcc39_by_stages(audio, layout, out)
        if out is None:
            out = torch.empty((layout.total_frames, width), dtype=torch.float32, device=audio.device)
        else:
            if (out.dtype != torch.float32 or not out.is_cuda or out.dim() != 2 or out.shape[0] != layout.total_frames
                    or out.shape[1] < width or out.stride(1) != 1):
                raise ValueError("out must be a float32 device tensor [total_frames, >=width] with unit column stride")
        code = capi.AUDIO_I16 if audio.dtype == torch.int16 else capi.AUDIO_F32
        if layout.device != self.plan.device:
            raise ValueError(f"layout on {layout.device}, plan on {self.plan.device}")
        with _on(self.plan.device, audio, out):
            capi.check(capi.load().mfcc_forward(self.plan._h, layout._h, _ptr(audio), code, _ptr(out), out.stride(0),
                                                _OUT_KINDS[kind], _stream_ptr()))
        return out

    def forward_sums(self, audio, layout, sums, out=None, workspace=None):
        """MFCC-39 rows as forward() writes them, and in the same launch the statistics of a corpus normalisation:
        sums (float64 device tensor [1 + 6 numcep], as cmvn_accumulate keeps it) += (rows, column sums, column
        sums of squares) of the rows written.  The writer adds its values up in float32 per tile of at most 296
        rows, the tiles are added in float64 in a fixed order: equal inputs give equal bits, and the sums agree
        with cmvn_accumulate over the same rows to about 1e-7 relative.  Plans of the specialised kernel only.
        ``workspace``: optional float32 device tensor that holds the per-tile sums (forward_sums_workspace()
        floats); allocated when missing."""
        if not audio.is_cuda or not audio.is_contiguous() or audio.dtype not in (torch.int16, torch.float32):
            raise ValueError("audio must be a contiguous int16 or float32 device tensor")
        if audio.numel() != layout.total_samples:
            raise ValueError(f"audio has {audio.numel()} samples, layout expects {layout.total_samples}")
        width = self.plan.out_width("mfcc39")
        if sums.dtype != torch.float64 or not sums.is_cuda or not sums.is_contiguous() or sums.numel() != 1 + 2 * width:
            raise ValueError(f"sums must be a contiguous float64 device tensor of {1 + 2 * width} elements")
        if out is None:
            out = torch.empty((layout.total_frames, width), dtype=torch.float32, device=audio.device)
        elif (out.dtype != torch.float32 or not out.is_cuda or out.dim() != 2 or out.shape[0] != layout.total_frames
              or out.shape[1] < width or out.stride(1) != 1):
            raise ValueError("out must be a float32 device tensor [total_frames, >=width] with unit column stride")
        lib = capi.load()
        need = capi.check(lib.mfcc_forward_sums_workspace(self.plan._h, layout._h)) // 4
        if workspace is None:
            workspace = torch.empty(need, dtype=torch.float32, device=audio.device)
        elif workspace.dtype != torch.float32 or not workspace.is_cuda or not workspace.is_contiguous() or workspace.numel() < need:
            raise ValueError(f"workspace must be a contiguous float32 device tensor of at least {need} elements")
        code = capi.AUDIO_I16 if audio.dtype == torch.int16 else capi.AUDIO_F32
        with _on(self.plan.device, audio, out):
            capi.check(lib.mfcc_forward_sums(self.plan._h, layout._h, _ptr(audio), code, _ptr(out), out.stride(0), _ptr(sums),
                                             _ptr(workspace), _stream_ptr()))
        return out

    def forward_sums_workspace(self, layout):
        """float32 elements of the workspace forward_sums() needs for this layout"""
        return capi.check(capi.load().mfcc_forward_sums_workspace(self.plan._h, layout._h)) // 4

    def _mfcc39_by_stages(self, audio, layout, out):
        """MFCC-39 for the rare plan whose delta rows do not fit the general kernel's LDS (mfcc_plan_supports):
        static rows in one launch, then delta and delta of delta per utterance (mfcc-self5.py:225-229)."""
        nc, pf = self.plan.numcep, int(self.plan.cfg.pad_fac)
        stat = self.forward(audio, layout, kind="static")
        if out is None:
            out = torch.empty((layout.total_frames, 3 * nc), dtype=torch.float32, device=audio.device)
        lib = capi.load()
        d = torch.empty_like(stat)
        dd = torch.empty_like(stat)
        with _on(self.plan.device, stat, out):
            st = _stream_ptr()
            fo = layout.frame_offsets
            for u in range(layout.n_utts):
                a, n = int(fo[u]), int(fo[u + 1] - fo[u])
                if n:
                    capi.check(lib.mfcc_delta(_ptr(stat[a:]), n, nc, pf, _ptr(d[a:]), st))
                    capi.check(lib.mfcc_delta(_ptr(d[a:]), n, nc, pf, _ptr(dd[a:]), st))
        out[:, :nc], out[:, nc:2 * nc], out[:, 2 * nc:3 * nc] = stat, d, dd
        return out

    def __call__(self, audio, lengths=None, out=None, kind="mfcc39"):
        t, _code, was_numpy = _audio_tensor(audio, self.plan.device)
        if lengths is None:
            lengths = [t.shape[-1]] * (t.shape[0] if t.dim() == 2 else 1)
        res = self.forward(t.reshape(-1), self.layout(lengths), out=out, kind=kind)
        return _ret(res, was_numpy)


def mfcc39(audio, lengths=None, fs=16000, frame_time=25, step_time=10, num_bins=26, low_freq=0, numcep=13,
           lift_factor=22, pad_fac=2, preemph=0.97, spectrum="power", window=None, floor_mel=False, out=None,
           kind="mfcc39", num_fft=None, frame_rule="pad", log_floor=0.0, high_freq=None, kernel="auto", keep_c0=False):
    """int16/float audio -> float32 [sum F, 39] = [c0=lnE, c1..c12 | delta | delta-delta]
    (the row layout of spectrograms(), mfcc-self5.py:181-186).  ``audio`` is one utterance
    [L], a dense batch [B, L], or a packed batch [sum L] with ``lengths``."""
    fe = MFCC.__new__(MFCC)
    fe.plan = get_plan(None if not isinstance(audio, torch.Tensor) or not audio.is_cuda else audio.device,
                       fs=fs, frame_time=frame_time, step_time=step_time, num_bins=num_bins, low_freq=low_freq,
                       numcep=numcep, lift_factor=lift_factor, pad_fac=pad_fac, preemph=preemph, spectrum=spectrum,
                       window=window, floor_mel=floor_mel, num_fft=num_fft, frame_rule=frame_rule, log_floor=log_floor,
                       high_freq=high_freq, kernel=kernel, keep_c0=keep_c0)
    fe._layouts = _layout_cache.setdefault(id(fe.plan), {})
    return fe(audio, lengths, out=out, kind=kind)


_layout_cache = {}


def spectrum257(audio, lengths=None, preemph=0.97, spectrum="power", window=None, log=False, out=None):
    """The 257-bin variant: winfft's per-frame |X|^2/512 (or |X|) rows, float32 [sum F, 257]
    (mfcc-self5.py:96-99); ``log=True`` returns their natural log."""
    return mfcc39(audio, lengths, preemph=preemph, spectrum=spectrum, window=window, out=out,
                  kind="logspec" if log else "spec")


def spectrum_stack(fe, audio, layout, framenum, out=None, log=True):
    """The "257 coefficients x N frames" input of the MLP/CNN scripts (MLP_5hidlayer_40.py:244-248)
    straight from audio: spectrum rows of ``fe`` (ln of them when ``log``) stacked over ``framenum``
    frames with edge replication per utterance -> float32 [sum F, framenum * (nfft/2+1)].  One launch
    for odd framenum up to 21 (each 257-bin row is computed once and written framenum times); larger
    windows go through ``forward`` + ``stack_context``."""
    if not audio.is_cuda or not audio.is_contiguous() or audio.dtype not in (torch.int16, torch.float32):
        raise ValueError("audio must be a contiguous int16 or float32 device tensor")
    if audio.numel() != layout.total_samples:
        raise ValueError(f"audio has {audio.numel()} samples, layout expects {layout.total_samples}")
    framenum = int(framenum)
    width = fe.plan.out_width("spec")
    shape = (layout.total_frames, framenum * width)
    if framenum % 2 != 1 or framenum > 21 or fe.plan.general or fe.plan.num_fft != 512:
        spec = fe.forward(audio, layout, kind="logspec" if log else "spec")
        return stack_context(spec, framenum, frame_counts=np.diff(layout.frame_offsets), out=out)
    if out is None:
        out = torch.empty(shape, dtype=torch.float32, device=audio.device)
    elif tuple(out.shape) != shape or out.dtype != torch.float32 or not out.is_contiguous() or out.data_ptr() % 16:
        raise ValueError(f"out must be a contiguous, 16-byte aligned float32 {shape} device tensor")
    code = capi.AUDIO_I16 if audio.dtype == torch.int16 else capi.AUDIO_F32
    with _on(fe.plan.device, audio, out):
        capi.check(capi.load().mfcc_forward_stacked(fe.plan._h, layout._h, _ptr(audio), code, framenum, _ptr(out),
                                                    capi.OUT_LOGSPEC if log else capi.OUT_SPEC, _stream_ptr()))
    return out


def mfcc39_stack(fe, audio, layout, framenum, out=None):
    """The "39 coefficients x N frames" input of the MLP scripts (coeffs = 39, e.g. framenum = 11,
    MLP_5hidlayer_40.py:244-248) straight from audio: the MFCC-39 rows of ``fe`` stacked over ``framenum``
    frames with edge replication per utterance (padUtterance + slidingWin, MLP_5hidlayer_40.py:70-105)
    -> float32 [sum F, framenum * 3 * numcep].  One launch for odd framenum 3 .. 21 on the specialised
    kernel (the rows never visit HBM on their own); otherwise ``forward`` + ``stack_context``."""
    if not audio.is_cuda or not audio.is_contiguous() or audio.dtype not in (torch.int16, torch.float32):
        raise ValueError("audio must be a contiguous int16 or float32 device tensor")
    if audio.numel() != layout.total_samples:
        raise ValueError(f"audio has {audio.numel()} samples, layout expects {layout.total_samples}")
    framenum = int(framenum)
    width = fe.plan.out_width("mfcc39")
    shape = (layout.total_frames, framenum * width)
    if framenum % 2 != 1 or framenum < 3 or framenum > 21 or fe.plan.general:
        rows = fe.forward(audio, layout, kind="mfcc39")
        return stack_context(rows, framenum, frame_counts=np.diff(layout.frame_offsets), out=out)
    if out is None:
        out = torch.empty(shape, dtype=torch.float32, device=audio.device)
    elif tuple(out.shape) != shape or out.dtype != torch.float32 or not out.is_contiguous() or out.data_ptr() % 16:
        raise ValueError(f"out must be a contiguous, 16-byte aligned float32 {shape} device tensor")
    code = capi.AUDIO_I16 if audio.dtype == torch.int16 else capi.AUDIO_F32
    with _on(fe.plan.device, audio, out):
        capi.check(capi.load().mfcc_forward_stacked(fe.plan._h, layout._h, _ptr(audio), code, framenum, _ptr(out),
                                                    capi.OUT_MFCC39, _stream_ptr()))
    return out


_frame_layouts = {}


def frames_layout(device, frame_counts):
    """Cached frame-count layout (building one uploads an offset table: keep it for repeated
    batch shapes)."""
    dev = _device(device)
    fc = np.ascontiguousarray(np.asarray(frame_counts, dtype=np.int64).reshape(-1))
    key = (dev.index, fc.tobytes())
    lay = _frame_layouts.get(key)
    if lay is None:
        if len(_frame_layouts) > 64:
            _frame_layouts.clear()
        lay = _frame_layouts[key] = Layout.from_frame_counts(dev, fc)
    return lay


def stack_context(feats, framenum, frame_counts=None, out=None, layout="frame_major", order=None):
    """[sum F, C] -> stacked windows with edge replication per utterance (padUtterance +
    slidingWin).  ``layout='frame_major'`` gives [n, framenum*C] with
    out[t, f*C + c] = feats[clamp(t + f - (N-1)//2), c] (the MLP vector, MLP_5hidlayer_40.py:70-105);
    ``layout='tile'`` gives [n, C, framenum], the [C, N] slice the CNN scripts keep
    (CNN_DenseNet40_6.py:98).  ``order`` (int64 rows of the packed array) selects and orders the
    output rows, e.g. the shuffled window order of CNN_DenseNet40_6.py:91-97; default all rows."""
    if layout not in ("frame_major", "tile"):
        raise ValueError("layout must be 'frame_major' or 'tile'")
    x, was_numpy = _as_rows(feats)
    t = _to_dev(x, torch.float32)
    if t.dim() != 2:
        raise ValueError("feats must be [frames, coeffs]")
    n, c = t.shape
    if frame_counts is None:
        frame_counts = [n]
    if int(np.sum(frame_counts)) != n:
        raise ValueError("frame_counts do not add up to the number of rows")
    lay = frames_layout(t.device, frame_counts)
    idx = None
    n_out = n
    if order is not None:
        o = np.asarray(order.cpu() if isinstance(order, torch.Tensor) else order, dtype=np.int64).reshape(-1)
        if len(o) and (o.min() < 0 or o.max() >= n):
            raise ValueError("order holds row numbers outside the feature array")
        idx = order.to(t.device).to(torch.int64).contiguous() if isinstance(order, torch.Tensor) else torch.from_numpy(o).to(t.device)
        n_out = len(o)
    shape = (n_out, framenum * c) if layout == "frame_major" else (n_out, c, framenum)
    if out is None:
        out = torch.empty(shape, dtype=torch.float32, device=t.device)
    elif tuple(out.shape) != shape or out.dtype != torch.float32 or not out.is_contiguous():
        raise ValueError(f"out must be a contiguous float32 {shape} device tensor")
    with _on(t.device, out, idx):
        capi.check(capi.load().mfcc_stack_context(
            lay._h, _ptr(t), t.stride(0), c, int(framenum), capi.STACK_FRAME_MAJOR if layout == "frame_major" else capi.STACK_TILE,
            _ptr(idx) if idx is not None else None, n_out, _ptr(out), _stream_ptr()))
    return _ret(out, was_numpy)


# --------------------------------------------------------------------------
# streaming
# --------------------------------------------------------------------------
class StreamingMFCC:
    """MFCC-39 for ``n_streams`` live streams that advance in lock step.

    ``push(chunk)`` takes [n_streams, n] samples (n <= max_chunk, any n, e.g. the 10 ms hops of
    a microphone queue as in get_mic_input_13.py:28-65) and returns the rows that became final,
    [n_streams, k, 39]; a row is final once the 4 frames after it are complete.  ``flush()``
    returns the remaining rows (zero-padded last frame, edge-replicated deltas) and starts a new
    utterance.  All rows of one utterance, concatenated, equal the whole-utterance result
    (main(), mfcc-self5.py:211-229) on the concatenated audio."""

    def __init__(self, variant="self5", n_streams=1, max_chunk=1600, dtype=torch.int16, device=None, **overrides):
        if dtype not in (torch.int16, torch.float32):
            raise ValueError("dtype must be torch.int16 or torch.float32")
        cfg = dict(VARIANTS[variant])
        cfg.update(overrides)
        self.plan = get_plan(device, **cfg)
        self.n_streams, self.max_chunk, self.dtype = int(n_streams), int(max_chunk), dtype
        self._h = C.c_void_p()
        with torch.cuda.device(self.plan.device):
            capi.check(capi.load().mfcc_stream_create(self.plan._h, self.n_streams, self.max_chunk,
                                                      capi.AUDIO_I16 if dtype == torch.int16 else capi.AUDIO_F32,
                                                      C.byref(self._h)))
        self.max_rows = capi.load().mfcc_stream_max_rows(self._h)
        self.width = self.plan.out_width("mfcc39")

    def _out(self):
        return torch.empty((self.n_streams, self.max_rows, self.width), dtype=torch.float32, device=self.plan.device)

    def push(self, chunk, out=None):
        was_numpy = not isinstance(chunk, torch.Tensor)
        t = torch.as_tensor(np.ascontiguousarray(chunk)) if was_numpy else chunk
        if t.dim() == 1:
            t = t.reshape(1, -1)
        if t.dim() != 2 or t.shape[0] != self.n_streams:
            raise ValueError(f"chunk must be [{self.n_streams}, n]")
        if t.shape[1] > self.max_chunk:
            raise ValueError(f"chunk of {t.shape[1]} samples, the stream was created for at most {self.max_chunk}")
        t = t.to(self.plan.device).to(self.dtype)
        if t.shape[1] > 0 and t.stride(1) != 1:
            t = t.contiguous()
        if out is None:
            out = self._out()
        k = C.c_int32(0)
        with _on(self.plan.device, out):
            # a single row may carry any stride in its size-1 dimension
            ld = max(t.stride(0), t.shape[1]) if self.n_streams == 1 else t.stride(0)
            capi.check(capi.load().mfcc_stream_push(self._h, _ptr(t), ld, t.shape[1],
                                                    _ptr(out), out.stride(0), out.stride(1), C.byref(k), _stream_ptr()))
        return _ret(out[:, :k.value], was_numpy)

    def flush(self, out=None, as_numpy=False):
        if out is None:
            out = self._out()
        k = C.c_int32(0)
        with _on(self.plan.device, out):
            capi.check(capi.load().mfcc_stream_flush(self._h, _ptr(out), out.stride(0), out.stride(1), C.byref(k),
                                                     _stream_ptr()))
        return _ret(out[:, :k.value], as_numpy)

    def reset(self):
        capi.check(capi.load().mfcc_stream_reset(self._h))

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                capi.load().mfcc_stream_destroy(h)
            except Exception:
                pass


# --------------------------------------------------------------------------
# CMVN
# --------------------------------------------------------------------------
def cmvn_accumulate(feats, sums=None, static=False):
    """Adds (n, sum x, sum x^2) of the rows of ``feats`` into the float64 device vector
    ``sums`` of length 1 + 2*D (allocated and zeroed when None).  The reduction order is fixed,
    so equal inputs give bit-identical sums.  ``static=True`` uses the entry point that keeps
    its partial sums in a fixed device array of the library (D <= 264) and never allocates."""
    if not feats.is_cuda or feats.dtype != torch.float32 or feats.dim() != 2 or feats.stride(1) != 1:
        raise ValueError("feats must be a float32 device tensor [rows, D] with unit column stride")
    n, d = feats.shape
    if sums is None:
        sums = torch.zeros(1 + 2 * d, dtype=torch.float64, device=feats.device)
    elif sums.dtype != torch.float64 or sums.numel() != 1 + 2 * d or not sums.is_contiguous():
        raise ValueError("sums must be a contiguous float64 vector of 1 + 2*D elements")
    lib = capi.load()
    fn = lib.mfcc_cmvn_accumulate_static if static else lib.mfcc_cmvn_accumulate
    with _on(feats.device, sums):
        capi.check(fn(_ptr(feats), feats.stride(0), n, d, _ptr(sums), _stream_ptr()))
    return sums


def cmvn_release(device=None):
    """Frees the partial-sum buffers cmvn_accumulate holds on ``device`` (one 16 MiB buffer per stream that
    has called it).  Waits for the device first."""
    dev = _device(device)
    with torch.cuda.device(dev):
        torch.cuda.synchronize(dev)
        capi.check(capi.load().mfcc_cmvn_release())


def cmvn_allreduce(sums, group=None):
    """Combine the CMVN statistics of all ranks (the only collective of the front-end, 1 + 2*D
    doubles per rank over RCCL): an all-gather followed by a sum in rank order on every rank, so
    the constants are bit-identical on all ranks and from run to run -- an all-reduce leaves the
    order of the additions to the collective's algorithm.  A no-op without a process group."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        # a host-side backend (gloo) gathers host copies; RCCL gathers the device vectors directly
        buf = sums.contiguous() if dist.get_backend(group) == "nccl" or not sums.is_cuda else sums.cpu()
        parts = [torch.empty_like(buf) for _ in range(dist.get_world_size(group))]
        dist.all_gather(parts, buf, group=group)
        total = parts[0].clone()
        for p in parts[1:]:
            total += p
        sums.copy_(total)
    return sums


def cmvn_constants(sums, ddof=0):
    """(mm, vv) float32 device vectors with mm = -mean, vv = 1/std so that
    (x + mm) * vv is the normalised feature (the form of anal_recog_nn_liste.m:180)."""
    d = (sums.numel() - 1) // 2
    n = sums[0]
    mean = sums[1:1 + d] / n
    var = (sums[1 + d:] - n * mean * mean) / (n - ddof)
    return (-mean).to(torch.float32), torch.rsqrt(var).to(torch.float32)


def cmvn_apply(feats, mm, vv):
    """In place: feats = (feats + mm) * vv."""
    n, d = feats.shape
    if mm.numel() != d or vv.numel() != d or mm.dtype != torch.float32 or vv.dtype != torch.float32:
        raise ValueError("mm and vv must be float32 vectors of D elements")
    with _on(feats.device, mm, vv):
        capi.check(capi.load().mfcc_cmvn_apply(_ptr(feats), feats.stride(0), n, d, _ptr(mm.contiguous()), _ptr(vv.contiguous()),
                                               _stream_ptr()))
    return feats


def shard_utterances(lengths, world_size, plan=None):
    """Contiguous blocks of utterances, balanced by frame count.  Returns the list of
    (begin, end) utterance ranges, one per rank."""
    lengths = np.asarray(lengths, dtype=np.int64).reshape(-1)
    size, step = (plan.frame_size, plan.frame_step) if plan is not None else (400, 160)
    frames = np.where(lengths <= size, 1, 1 + -(-(lengths - size) // step))
    csum = np.concatenate([[0], np.cumsum(frames)])
    total = csum[-1]
    bounds = [0]
    for r in range(1, world_size):
        target = total * r / world_size
        b = int(np.searchsorted(csum, target, side="left"))
        b = min(max(b, bounds[-1]), len(lengths))
        bounds.append(b)
    bounds.append(len(lengths))
    return [(bounds[r], bounds[r + 1]) for r in range(world_size)]


# --------------------------------------------------------------------------
# the reference's function names (mfcc-self5.py:32-186)
# --------------------------------------------------------------------------
def loadrawdata(filename):
    """Headerless little-endian int16 file -> int16 array (mfcc-self5.py:32-36)."""
    with open(filename, "rb") as fid:
        return np.fromfile(fid, dtype=np.int16)


def preemphasis(signal, coeff=0.97):
    """mfcc-self5.py:66-68."""
    t, code, was_numpy = _audio_tensor(signal)
    t = t.reshape(-1)
    out = torch.empty(t.numel(), dtype=torch.float32, device=t.device)
    with _on(t.device, out):
        capi.check(capi.load().mfcc_preemphasis(_ptr(t), code, t.numel(), float(coeff), _ptr(out), _stream_ptr()))
    return _ret(out, was_numpy)


def winfft(audio, fs, frame_time=25, step_time=10, window=np.hamming):
    """mfcc-self5.py:70-100.  Returns (magfft, powfft, win, num_fft); the two spectra are
    [F, num_fft//2+1] arrays that iterate frame by frame like the reference's lists.
    ``window`` is evaluated and returned but not applied, as in the reference (:78,:95)."""
    t, _code, was_numpy = _audio_tensor(audio)
    t = t.reshape(-1)
    common = dict(fs=fs, frame_time=frame_time, step_time=step_time, preemph=None)
    fe_p, fe_m = MFCC.__new__(MFCC), MFCC.__new__(MFCC)
    fe_p.plan, fe_m.plan = get_plan(t.device, spectrum="power", **common), get_plan(t.device, spectrum="mag", **common)
    lay = Layout(fe_p.plan, [t.numel()])
    powfft = fe_p.forward(t, lay, kind="spec")
    magfft = fe_m.forward(t, lay, kind="spec")
    win = window(fe_p.plan.frame_size)
    return _ret(magfft, was_numpy), _ret(powfft, was_numpy), win, np.int64(fe_p.plan.num_fft)


def melbanks(fs, numbins=26, lowfreq=0, num_fft=None):
    """mfcc-self5.py:102-117 -> float64 [numbins, num_fft//2+1] (host constant table)."""
    if num_fft is None:
        raise TypeError("num_fft is required (the reference multiplies by num_fft+1, mfcc-self5.py:109)")
    out = np.zeros((int(numbins), int(num_fft) // 2 + 1), dtype=np.float64)
    capi.check(capi.load().mfcc_melbank_host(float(fs), int(numbins), float(lowfreq), int(num_fft),
                                             out.ctypes.data_as(C.POINTER(C.c_double))))
    return out


def melspec(powfft_list, melbank):
    """mfcc-self5.py:119-134 -> (melspec [F, nmel] = ln(spec @ melbank.T) unfloored, energy [F])."""
    x, was_numpy = _as_rows(powfft_list)
    spec = _to_dev(x, torch.float32)
    bank = _to_dev(melbank, torch.float32, spec.device)
    n, k = spec.shape
    nmel = bank.shape[0]
    if bank.shape[1] != k:
        raise ValueError(f"shapes {tuple(spec.shape)} and {tuple(bank.T.shape)} not aligned")
    logmel = torch.empty((n, nmel), dtype=torch.float32, device=spec.device)
    loge = torch.empty(n, dtype=torch.float32, device=spec.device)
    with _on(spec.device, bank):
        capi.check(capi.load().mfcc_melspec(_ptr(spec), n, k, _ptr(bank), nmel, 0, _ptr(logmel), _ptr(loge), _stream_ptr()))
    return _ret(logmel, was_numpy), _ret(loge, was_numpy)


def mfcc(melspec_list, numcep=12):
    """mfcc-self5.py:136-142: DCT-II, norm='ortho', first numcep coefficients per frame."""
    x, was_numpy = _as_rows(melspec_list)
    lm = _to_dev(x, torch.float32)
    n, nmel = lm.shape
    out = torch.empty((n, int(numcep)), dtype=torch.float32, device=lm.device)
    with _on(lm.device):
        capi.check(capi.load().mfcc_dct(_ptr(lm), n, nmel, int(numcep), _ptr(out), _stream_ptr()))
    return _ret(out, was_numpy)


def dctopt(mfcc_list, lift_factor=22):
    """mfcc-self5.py:144-156: sinusoidal liftering (identity for lift_factor <= 0)."""
    x, was_numpy = _as_rows(mfcc_list)
    c = _to_dev(x, torch.float32)
    n, k = c.shape
    out = torch.empty_like(c)
    with _on(c.device):
        capi.check(capi.load().mfcc_lifter(_ptr(c), n, k, float(lift_factor), _ptr(out), _stream_ptr()))
    return _ret(out, was_numpy)


def staticfeatures(energy_list, mfcclift_list):
    """mfcc-self5.py:158-167: coefficient 0 replaced by the log energy.  Unlike the reference
    the input is not modified in place."""
    x, was_numpy = _as_rows(mfcclift_list)
    c = _to_dev(x, torch.float32)
    e = _to_dev(np.asarray(energy_list) if not isinstance(energy_list, torch.Tensor) else energy_list, torch.float32,
                c.device).reshape(-1, 1)
    n, k = c.shape
    out = torch.empty_like(c)
    lib = capi.load()
    with _on(c.device, e):
        capi.check(lib.mfcc_copy_cols(_ptr(c), c.stride(0), _ptr(out), out.stride(0), n, k, _stream_ptr()))
        capi.check(lib.mfcc_copy_cols(_ptr(e), 1, _ptr(out), out.stride(0), n, 1, _stream_ptr()))
    return _ret(out, was_numpy)


def delta(features, pad_fac=2):
    """mfcc-self5.py:169-179."""
    x, was_numpy = _as_rows(features)
    f = _to_dev(x, torch.float32)
    n, k = f.shape
    out = torch.empty_like(f)
    with _on(f.device):
        capi.check(capi.load().mfcc_delta(_ptr(f), n, k, int(pad_fac), _ptr(out), _stream_ptr()))
    return _ret(out, was_numpy)


def spectrograms(cepstrum, delta_features, delta_delta_features):
    """mfcc-self5.py:181-186: column concatenation -> [F, 39]."""
    parts, was_numpy = [], True
    for p in (cepstrum, delta_features, delta_delta_features):
        x, wn = _as_rows(p)
        was_numpy = was_numpy and wn
        parts.append(_to_dev(x, torch.float32, parts[0].device if parts else None))
    n = parts[0].shape[0]
    width = sum(p.shape[1] for p in parts)
    out = torch.empty((n, width), dtype=torch.float32, device=parts[0].device)
    col = 0
    lib = capi.load()
    for p in parts:
        dst = C.c_void_p(out.data_ptr() + 4 * col)
        with _on(out.device, p):
            capi.check(lib.mfcc_copy_cols(_ptr(p), p.stride(0), dst, out.stride(0), n, p.shape[1], _stream_ptr()))
        col += p.shape[1]
    return _ret(out, was_numpy)


# --------------------------------------------------------------------------
# .pat container and the stacking functions of MLP_5hidlayer_40.py:61-105
# --------------------------------------------------------------------------
coeffs = 39  # module-level global the reference's loadPatData / slidingWin read (MLP_5hidlayer_40.py:245)


def loadPatData(filename, headerbytes=12, ncoeffs=None):
    """12-byte HTK-style header {int32 nFrames, int32 sampPeriod, int16 sampSize, int16
    parmKind} + float32 [nFrames, C]; returned transposed as [C, F] like the reference
    (MLP_5hidlayer_40.py:61-68)."""
    with open(filename, "rb") as fid:
        raw = fid.read()
    nframes, _period, samp_size, _kind = struct.unpack("<iihh", raw[:12])
    c = int(ncoeffs) if ncoeffs is not None else (samp_size // 4 if samp_size > 0 else coeffs)
    data = np.frombuffer(raw, dtype="<f4", offset=headerbytes, count=nframes * c)
    return data.reshape(nframes, c).T


def writePatData(filename, feats, samp_period=100000, parm_kind=838):
    """feats [F, C] float32 -> file readable by loadPatData (parmKind 838 = MFCC_E_D_A,
    10 ms frame period, as in the reference's 39pat files)."""
    feats = np.ascontiguousarray(np.asarray(feats, dtype="<f4"))
    nframes, c = feats.shape
    with open(filename, "wb") as fid:
        fid.write(struct.pack("<iihh", nframes, samp_period, c * 4, parm_kind))
        fid.write(feats.tobytes())


def padUtterance(framenum, labelfile, datafile):
    """MLP_5hidlayer_40.py:70-87: replicate the first/last frame (and label) (N-1)/2 times.
    datafile is [C, F]; returns (paddedlabel, paddeddata [C, F + N - 1])."""
    labelfile = np.asarray(labelfile)
    x, was_numpy = _as_rows(datafile)
    nframes = x.shape[1]
    if labelfile.shape[0] != nframes:
        labelfile = labelfile[0:-1]
    half = int((framenum - 1) / 2)
    paddedlabel = np.concatenate((np.tile(int(labelfile[0]), half), labelfile, np.tile(int(labelfile[-1]), half)))
    feats = _to_dev(x, torch.float32).t().contiguous()  # [F, C]
    # edge replication expressed as a 1-frame "stack" of the clamped rows: row t of the
    # padded data is feats[clamp(t - half)]
    idx = torch.clamp(torch.arange(-half, nframes + half, device=feats.device), 0, nframes - 1)
    padded = feats.index_select(0, idx).t()
    return paddedlabel, _ret(padded.contiguous(), was_numpy)


def slidingWin(framenum, paddedlabel, paddeddata, randomint=-11):
    """MLP_5hidlayer_40.py:89-105: every window of N consecutive padded frames, flattened
    frame-major ([C,N] slice reshaped in Fortran order) -> (labellist, datalist).  The
    datalist is a [F, N*C] array that iterates vector by vector.  ``randomint`` is accepted
    and ignored exactly as the reference ignores the permutation it computes (:94-96)."""
    paddedlabel = np.asarray(paddedlabel)
    x, was_numpy = _as_rows(paddeddata)
    half = int((framenum - 1) / 2)
    nwin = paddedlabel.shape[0] - framenum + 1
    labellist = [paddedlabel[pos + half] for pos in range(nwin)]
    padded = _to_dev(x, torch.float32).t().contiguous()  # [Fp, C]
    stacked = stack_context(padded, framenum)  # clamps only in rows outside [half, Fp-half)
    data = stacked[half:half + nwin]
    return labellist, _ret(data, was_numpy)
