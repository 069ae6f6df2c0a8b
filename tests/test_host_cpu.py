"""CPU-only checks of the product's host side: the shared library loads, exports every
symbol declared in include/mfcc_amd.h, and its host-side helpers reproduce the reference's
constants.  No kernel is launched here."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import ROOT

capi = pytest.importorskip("mfcc_amd._capi")


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(capi.LIB_PATH):
        subprocess.check_call(["make", "-C", ROOT, "all"])
    return capi.load()


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "mfcc_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mfcc_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound(lib):
    names = declared_symbols()
    assert len(names) >= 20
    for name in names:
        assert hasattr(lib, name), f"{name} declared in mfcc_amd.h but not exported"
        assert name in capi.SIGNATURES, f"{name} has no ctypes signature"
    assert sorted(capi.SIGNATURES) == names


def test_config_struct_matches_header(lib):
    cfg = capi.MfccConfig()
    lib.mfcc_config_default(C.byref(cfg))
    assert (cfg.fs, cfg.frame_time, cfg.step_time, cfg.num_bins, cfg.low_freq) == (16000, 25, 10, 26, 0)
    assert (cfg.numcep, cfg.lift_factor, cfg.pad_fac, cfg.use_preemph, cfg.preemph) == (13, 22, 2, 1, 0.97)
    assert (cfg.spectrum, cfg.window, cfg.floor_mel) == (0, 0, 0)


def test_melbank_host_is_bit_exact(lib, golden):
    g = golden("ldc93s1.npz")
    s = golden("stages.npz")
    for args, ref in [((16000.0, 26, 0.0, 512), g["melbank"]), ((16000.0, 40, 0.0, 512), s["melbank_40"]),
                      ((16000.0, 26, 300.0, 512), s["melbank_26_low300"]), ((8000.0, 26, 0.0, 256), s["melbank_8k_26_256"])]:
        out = np.zeros(ref.shape)
        assert lib.mfcc_melbank_host(*args, out.ctypes.data_as(C.POINTER(C.c_double))) == 0
        assert np.array_equal(out, ref)


def test_frame_geometry_and_counts(lib):
    a, b, c = C.c_int32(), C.c_int32(), C.c_int32()
    assert lib.mfcc_frame_geometry(16000.0, 25.0, 10.0, C.byref(a), C.byref(b), C.byref(c)) == 0
    assert (a.value, b.value, c.value) == (400, 160, 512)
    assert lib.mfcc_frame_geometry(8000.0, 25.0, 10.0, C.byref(a), C.byref(b), C.byref(c)) == 0
    assert (a.value, b.value, c.value) == (200, 80, 256)
    for length, frames in [(1, 1), (399, 1), (400, 1), (401, 2), (560, 2), (561, 3), (4000, 24), (48000, 299), (46797, 291)]:
        assert lib.mfcc_frames_for_length(length, 400, 160) == frames
    assert lib.mfcc_frames_for_length(-1, 400, 160) == capi.ERR_INVALID
    assert b"bad argument" in lib.mfcc_last_error()


def test_argument_errors_without_gpu(lib):
    assert lib.mfcc_plan_create(None, 0, None) == capi.ERR_INVALID
    cfg = capi.MfccConfig()
    lib.mfcc_config_default(C.byref(cfg))
    cfg.fs = 192000.0   # 4800-sample frames would need an 8192-point FFT: refused before any device call
    h = C.c_void_p()
    assert lib.mfcc_plan_create(C.byref(cfg), 0, C.byref(h)) == capi.ERR_UNSUPPORTED
    assert b"FFT lengths" in lib.mfcc_last_error()
    cfg.fs, cfg.pad_fac = 16000.0, 0
    assert lib.mfcc_plan_create(C.byref(cfg), 0, C.byref(h)) == capi.ERR_UNSUPPORTED
    cfg.pad_fac, cfg.kernel = 2, 7
    assert lib.mfcc_plan_create(C.byref(cfg), 0, C.byref(h)) == capi.ERR_INVALID
    assert lib.mfcc_forward(None, None, None, 0, None, 39, 0, None) == capi.ERR_INVALID
    assert lib.mfcc_delta(None, -1, 13, 2, None, None) == capi.ERR_INVALID


@pytest.mark.parametrize("frame_size", [400, 512, 7])
def test_lane_level_fft_emulation(frame_size):
    exe = os.path.join(ROOT, "tests", "host_sim", "fft_sim")
    subprocess.check_call(["make", "-C", ROOT, "host_sim"])
    res = subprocess.run([exe, str(frame_size)], capture_output=True, text=True)
    assert res.returncode == 0, res.stdout + res.stderr
    assert float(res.stdout.split()[-1]) < 2e-6


def test_product_never_imports_oracle():
    for dirpath, _d, files in os.walk(os.path.join(ROOT, "mfcc_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert "oracle" not in text, f"{f} mentions the oracle"


def test_shard_utterances_balances_frames():
    torch = pytest.importorskip("torch")  # noqa: F841
    from mfcc_amd.frontend import shard_utterances
    rng = np.random.default_rng(1)
    lengths = rng.integers(400, 160000, size=1000)
    for world in (1, 2, 4, 8):
        parts = shard_utterances(lengths, world)
        assert parts[0][0] == 0 and parts[-1][1] == 1000
        assert all(parts[i][1] == parts[i + 1][0] for i in range(world - 1))
        frames = [sum(1 + -(-(int(n) - 400) // 160) for n in lengths[a:b]) for a, b in parts]
        assert max(frames) - min(frames) <= 2 * 1000


def test_dataset_helpers_follow_the_reference(golden):
    """Host-only helpers of the .pat pipeline (MLP_5hidlayer_40.py:107-142)."""
    from mfcc_amd import datasets as D
    v = D.createRandomvec(3, 10)
    np.random.seed(3)
    ref = np.arange(10)
    np.random.shuffle(ref)
    assert np.array_equal(v, ref)
    assert D.randomShuffleData(3, list("abcdefghij")) == [list("abcdefghij")[i] for i in ref]
    lab = D.createMinibatchLabel(3, 5, [1.0, 5.0, 2.0])
    assert lab.shape == (3, 5) and lab[0, 0] == 1 and lab[1, 4] == 1 and lab[2, 1] == 1 and lab.sum() == 3
    x = np.arange(2 * 39, dtype=np.float32).reshape(2, 39)
    y = D.energy_last(x)
    assert np.array_equal(y[0, :13], np.r_[1:13, 0]) and np.array_equal(y[0, 13:26], np.r_[14:26, 13])


def test_config_struct_matches_the_binding():
    """The ctypes mirror of struct mfcc_config must have the library's size and defaults."""
    import ctypes as C
    from mfcc_amd import _capi as capi
    lib = capi.load()
    assert lib.mfcc_config_size() == C.sizeof(capi.MfccConfig)
    cfg = capi.MfccConfig()
    lib.mfcc_config_default(C.byref(cfg))
    assert (cfg.fs, cfg.frame_time, cfg.step_time, cfg.num_bins, cfg.numcep, cfg.lift_factor, cfg.pad_fac) == (16000, 25, 10, 26, 13, 22, 2)
    assert (cfg.use_preemph, cfg.preemph, cfg.spectrum, cfg.window, cfg.floor_mel, cfg.num_fft, cfg.frame_rule, cfg.log_floor) == \
        (1, 0.97, capi.SPECTRUM_POWER, capi.WINDOW_NONE, 0, 0, 0, 0.0)
