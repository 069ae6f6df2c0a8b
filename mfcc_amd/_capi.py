"""ctypes binding of libmfcc_amd.so (include/mfcc_amd.h).

The library is the product: there is no Python or CPU fallback.  Importing this
module without the built shared object raises ImportError with the build
command; calling into it without a GPU raises from the HIP runtime.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libmfcc_amd.so")

OK = 0
ERR_INVALID, ERR_UNSUPPORTED, ERR_HIP, ERR_NOMEM = -1, -2, -3, -4
AUDIO_I16, AUDIO_F32 = 0, 1
SPECTRUM_POWER, SPECTRUM_MAG = 0, 1
WINDOW_NONE, WINDOW_HAMMING = 0, 1
OUT_MFCC39, OUT_STATIC, OUT_LOGMEL, OUT_SPEC, OUT_LOGSPEC = 0, 1, 2, 3, 4
STACK_FRAME_MAJOR, STACK_TILE = 0, 1
KERNEL_AUTO, KERNEL_GENERAL = 0, 1
CMVN_STATIC_MAX_COLS = 264


class MfccConfig(C.Structure):
    _fields_ = [
        ("fs", C.c_double),
        ("frame_time", C.c_double),
        ("step_time", C.c_double),
        ("num_bins", C.c_int32),
        ("low_freq", C.c_double),
        ("numcep", C.c_int32),
        ("lift_factor", C.c_double),
        ("pad_fac", C.c_int32),
        ("use_preemph", C.c_int32),
        ("preemph", C.c_double),
        ("spectrum", C.c_int32),
        ("window", C.c_int32),
        ("floor_mel", C.c_int32),
        ("num_fft", C.c_int32),
        ("frame_rule", C.c_int32),
        ("log_floor", C.c_double),
        ("high_freq", C.c_double),
        ("kernel", C.c_int32),
        ("keep_c0", C.c_int32),
    ]


class MfccError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libmfcc_amd error {code}: {msg}")
        self.code = code


_vp, _i32, _i64, _dbl = C.c_void_p, C.c_int32, C.c_int64, C.c_double
_i32p, _i64p, _dblp = C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_double)

# name -> (restype, argtypes); every symbol declared in include/mfcc_amd.h
SIGNATURES = {
    "mfcc_version": (C.c_int, []),
    "mfcc_last_error": (C.c_char_p, []),
    "mfcc_config_default": (None, [C.POINTER(MfccConfig)]),
    "mfcc_config_size": (_i32, []),
    "mfcc_melbank_host": (C.c_int, [_dbl, _i32, _dbl, _i32, _dblp]),
    "mfcc_frame_geometry": (C.c_int, [_dbl, _dbl, _dbl, _i32p, _i32p, _i32p]),
    "mfcc_frames_for_length": (_i64, [_i64, _i32, _i32]),
    "mfcc_plan_create": (C.c_int, [C.POINTER(MfccConfig), C.c_int, C.POINTER(_vp)]),
    "mfcc_plan_destroy": (C.c_int, [_vp]),
    "mfcc_plan_info": (C.c_int, [_vp, _i32p, _i32p, _i32p, _i32p, _i32p]),
    "mfcc_plan_kernel": (_i32, [_vp]),
    "mfcc_plan_supports": (_i32, [_vp, _i32]),
    "mfcc_layout_create": (C.c_int, [_vp, _i64p, _i64, C.POINTER(_vp)]),
    "mfcc_layout_create_frames": (C.c_int, [C.c_int, _i64p, _i64, C.POINTER(_vp)]),
    "mfcc_layout_destroy": (C.c_int, [_vp]),
    "mfcc_layout_info": (C.c_int, [_vp, _i64p, _i64p, _i64p]),
    "mfcc_layout_offsets_host": (C.c_int, [_vp, _i64p, _i64p]),
    "mfcc_forward": (C.c_int, [_vp, _vp, _vp, _i32, _vp, _i64, _i32, _vp]),
    "mfcc_forward_sums_workspace": (_i64, [_vp, _vp]),
    "mfcc_forward_sums": (C.c_int, [_vp, _vp, _vp, _i32, _vp, _i64, _vp, _vp, _vp]),
    "mfcc_out_width": (_i32, [_vp, _i32]),
    "mfcc_forward_stacked": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _vp, _i32, _vp]),
    "mfcc_preemphasis": (C.c_int, [_vp, _i32, _i64, _dbl, _vp, _vp]),
    "mfcc_melspec": (C.c_int, [_vp, _i64, _i32, _vp, _i32, _i32, _vp, _vp, _vp]),
    "mfcc_dct": (C.c_int, [_vp, _i64, _i32, _i32, _vp, _vp]),
    "mfcc_lifter": (C.c_int, [_vp, _i64, _i32, _dbl, _vp, _vp]),
    "mfcc_delta": (C.c_int, [_vp, _i64, _i32, _i32, _vp, _vp]),
    "mfcc_fir": (C.c_int, [_vp, _vp, _i32, C.POINTER(C.c_float), _i32, _i32, _vp, _vp]),
    "mfcc_copy_cols": (C.c_int, [_vp, _i64, _vp, _i64, _i64, _i32, _vp]),
    "mfcc_stack_context": (C.c_int, [_vp, _vp, _i64, _i32, _i32, _i32, _vp, _i64, _vp, _vp]),
    "mfcc_cmvn_accumulate": (C.c_int, [_vp, _i64, _i64, _i32, _vp, _vp]),
    "mfcc_cmvn_accumulate_static": (C.c_int, [_vp, _i64, _i64, _i32, _vp, _vp]),
    "mfcc_cmvn_release": (C.c_int, []),
    "mfcc_cmvn_apply": (C.c_int, [_vp, _i64, _i64, _i32, _vp, _vp, _vp]),
    "mfcc_utt_scalar_norm": (C.c_int, [_vp, _vp, _i64, _i32, _vp]),
    "mfcc_pad_batch": (C.c_int, [_vp, _vp, _i64, _i32, _i32, C.c_float, _i32, _i32, _vp, _vp]),
    "mfcc_stream_create": (C.c_int, [_vp, _i32, _i32, _i32, C.POINTER(_vp)]),
    "mfcc_stream_destroy": (C.c_int, [_vp]),
    "mfcc_stream_reset": (C.c_int, [_vp]),
    "mfcc_stream_max_rows": (_i32, [_vp]),
    "mfcc_stream_push": (C.c_int, [_vp, _vp, _i64, _i32, _vp, _i64, _i64, _i32p, _vp]),
    "mfcc_stream_flush": (C.c_int, [_vp, _vp, _i64, _i64, _i32p, _vp]),
}

_lib = None


def load():
    """Load the shared library once.  torch must already be imported by the caller
    that owns device memory: both link the HIP runtime by SONAME, so the runtime
    torch loaded is the one this library binds to."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `make` (hipcc --offload-arch=gfx950) "
            "or python -c 'import __graft_entry__ as g; g.build()'. There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if lib.mfcc_config_size() != C.sizeof(MfccConfig):
        raise ImportError(f"{LIB_PATH}: struct mfcc_config has {lib.mfcc_config_size()} bytes, this binding expects "
                          f"{C.sizeof(MfccConfig)}: rebuild the library (make)")
    _lib = lib
    return lib


def check(rc):
    if rc is not None and rc < 0:
        raise MfccError(rc, load().mfcc_last_error().decode("utf-8", "replace"))
    return rc
