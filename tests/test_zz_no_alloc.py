"""mfcc_cmvn_accumulate_static -- the entry point that never allocates (its partial sums live in a
fixed device array of the library) -- against mfcc_cmvn_accumulate, through the C ABI, on two
calls in a row, the second with a smaller input, bit for bit."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _call(fn, feats, sums):
    from mfcc_amd import _capi as capi
    capi.check(fn(C.c_void_p(feats.data_ptr()), feats.stride(0), feats.shape[0], feats.shape[1], C.c_void_p(sums.data_ptr()),
                  C.c_void_p(torch.cuda.current_stream().cuda_stream)))


def test_static_scratch_entry_matches_allocating_entry():
    assert torch.cuda.is_available()
    from mfcc_amd import _capi as capi
    lib = capi.load()
    gen = torch.Generator(device="cuda").manual_seed(5)
    big = torch.randn(700001, 39, generator=gen, device="cuda") * 7 + 2        # 1024 partial blocks in use
    small = torch.randn(1537, 257, generator=gen, device="cuda") * 3 - 1       # fewer blocks, wider rows
    for feats in (big, small):
        d = feats.shape[1]
        ref = torch.zeros(1 + 2 * d, dtype=torch.float64, device="cuda")
        got = torch.full((1 + 2 * d,), 0.0, dtype=torch.float64, device="cuda")
        _call(lib.mfcc_cmvn_accumulate, feats, ref)
        _call(lib.mfcc_cmvn_accumulate_static, feats, got)
        torch.cuda.synchronize()
        assert torch.equal(got, ref)
        assert got[0].item() == feats.shape[0]
        x = feats.double()
        assert torch.allclose(got[1:1 + d], x.sum(0), rtol=1e-12, atol=1e-6)
        assert torch.allclose(got[1 + d:], (x * x).sum(0), rtol=1e-12)
    # accumulation into non-zero sums and a padded row stride
    padded = torch.randn(5000, 48, generator=gen, device="cuda")
    view = padded[:, :39]
    ref = torch.ones(79, dtype=torch.float64, device="cuda")
    got = torch.ones(79, dtype=torch.float64, device="cuda")
    _call(lib.mfcc_cmvn_accumulate, view, ref)
    _call(lib.mfcc_cmvn_accumulate_static, view, got)
    assert torch.equal(got, ref)
    # beyond the static scratch: a clear error, nothing launched
    wide = torch.zeros(4, capi.CMVN_STATIC_MAX_COLS + 1, device="cuda")
    sums = torch.zeros(1 + 2 * wide.shape[1], dtype=torch.float64, device="cuda")
    rc = lib.mfcc_cmvn_accumulate_static(C.c_void_p(wide.data_ptr()), wide.stride(0), 4, wide.shape[1], C.c_void_p(sums.data_ptr()), None)
    assert rc == capi.ERR_UNSUPPORTED and b"static scratch" in lib.mfcc_last_error()
    assert sums.abs().sum().item() == 0
    # the Python wrapper reaches the same entry point
    import mfcc_amd
    assert torch.equal(mfcc_amd.cmvn_accumulate(big, static=True), mfcc_amd.cmvn_accumulate(big))
