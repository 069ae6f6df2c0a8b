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


def test_static_entry_allocates_nothing_and_captures():
    """The static entry point under stream capture (an allocation there would fail the capture), replayed on
    new data; device memory in use does not move over a run of calls; calls from two streams take turns on
    the shared array (each waits for the event of the call before it) and both give the allocating entry's bits."""
    from mfcc_amd import _capi as capi
    lib = capi.load()
    gen = torch.Generator(device="cuda").manual_seed(9)
    feats = torch.randn(300007, 39, generator=gen, device="cuda")
    ref = torch.zeros(79, dtype=torch.float64, device="cuda")
    _call(lib.mfcc_cmvn_accumulate, feats, ref)
    got = torch.zeros(79, dtype=torch.float64, device="cuda")
    _call(lib.mfcc_cmvn_accumulate_static, feats, got)          # address resolved, event created: set-up is over
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    for _ in range(20):
        got.zero_()
        _call(lib.mfcc_cmvn_accumulate_static, feats, got)
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] == free0
    assert torch.equal(got, ref)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        got.zero_()
        _call(lib.mfcc_cmvn_accumulate_static, feats, got)
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        _call(lib.mfcc_cmvn_accumulate_static, feats, got)
    got.zero_()
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(got, ref)
    feats2 = torch.randn(300007, 39, generator=gen, device="cuda") * 3
    ref2 = torch.zeros(79, dtype=torch.float64, device="cuda")
    _call(lib.mfcc_cmvn_accumulate, feats2, ref2)
    feats.copy_(feats2)
    got.zero_()
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(got, ref2)
    # eager calls after a capture still work (the event was never recorded into the graph), from two streams at once
    a = torch.randn(200003, 39, generator=gen, device="cuda")
    b = torch.randn(150001, 257, generator=gen, device="cuda")
    ra, rb = torch.zeros(79, dtype=torch.float64, device="cuda"), torch.zeros(515, dtype=torch.float64, device="cuda")
    _call(lib.mfcc_cmvn_accumulate, a, ra)
    _call(lib.mfcc_cmvn_accumulate, b, rb)
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    for _ in range(5):
        ga, gb = torch.zeros_like(ra), torch.zeros_like(rb)
        torch.cuda.synchronize()
        with torch.cuda.stream(s1):
            _call(lib.mfcc_cmvn_accumulate_static, a, ga)
        with torch.cuda.stream(s2):
            _call(lib.mfcc_cmvn_accumulate_static, b, gb)
        torch.cuda.synchronize()
        assert torch.equal(ga, ra) and torch.equal(gb, rb)


def test_workspace_of_the_allocating_entry_is_released():
    import mfcc_amd
    x = torch.randn(4096, 39, device="cuda")
    mfcc_amd.cmvn_accumulate(x)
    torch.cuda.synchronize()
    held = torch.cuda.mem_get_info()[0]
    mfcc_amd.cmvn_release()
    assert torch.cuda.mem_get_info()[0] >= held + (15 << 20)     # the 16 MiB buffer of this stream went back
    assert torch.equal(mfcc_amd.cmvn_accumulate(x), mfcc_amd.cmvn_accumulate(x, static=True))
