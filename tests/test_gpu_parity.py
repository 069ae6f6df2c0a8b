"""GPU parity tests: the HIP path (through the C ABI) against the golden vectors produced by
the reference's own functions and against the CPU oracle on seeded inputs.

Tolerances (float32 kernels vs the float64 reference):
  * ln-mel / ln-energy: 1e-4 RELATIVE, |gpu - ref| <= 1e-4 * max(|ref|, 1).  The floor of 1
    on the denominator matters only where ln-mel is within +-1 of zero (mel energy ~ 1, a
    few entries of near-silent frames in LDC93S1).  There a pure relative bound asks for
    the mel energy to ~3e-6, which no float32 FFT delivers when the frame holds content
    60 dB stronger at other frequencies: the rounding noise of a float32 FFT is ~1e-7 of the
    frame's RMS spectrum (scipy.fft.rfft in float32 gives 1.2e-4 on the same entry; numpy's
    float32 FFT is computed in double internally and is not a float32 datum).  The pure
    relative figure is still measured, printed and bounded: >= 99.9 % of entries within 1e-4 and
    no entry beyond 3e-4 (the worst observed is 2.4e-4, LDC93S1 frame 19 band 1);
  * cepstra, delta, delta-delta (all 39 columns): 1e-3 ABSOLUTE;
  * integer / copy operations (stacking, .pat round trip): bit exact.
"""
import os
import sys

import numpy as np
import pytest
import torch

from oracle import mfcc_oracle as O

pytestmark = pytest.mark.gpu

LOGMEL_RTOL = 1e-4
LOGMEL_PURE_CAP = 3e-4   # entries where ln-mel crosses zero; worst observed 2.4e-4
CEP_ATOL = 1e-3
VARIANTS = ("self5", "self15")


@pytest.fixture(scope="module")
def M():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    import mfcc_amd
    return mfcc_amd


def assert_logmel(gpu, ref, what):
    gpu = np.asarray(gpu, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    assert gpu.shape == ref.shape, what
    fin = np.isfinite(ref)
    assert np.array_equal(np.isfinite(gpu), fin), f"{what}: inf/nan pattern differs"
    assert np.array_equal(gpu[~fin], ref[~fin]) or np.array_equal(np.sign(gpu[~fin]), np.sign(ref[~fin]))
    err = np.abs(gpu[fin] - ref[fin])
    rel = err / np.maximum(np.abs(ref[fin]), 1.0)
    assert rel.max(initial=0) <= LOGMEL_RTOL, f"{what}: max rel err {rel.max():.3e} at |ref|={np.abs(ref[fin])[rel.argmax()]:.4g}"
    pure = err / np.maximum(np.abs(ref[fin]), 1e-300)
    if pure.size:
        print(f"[ln-mel] {what}: worst pure relative {pure.max():.3e} (|ref| = {np.abs(ref[fin])[pure.argmax()]:.4g}), "
              f"{int((pure > LOGMEL_RTOL).sum())} of {pure.size} beyond {LOGMEL_RTOL:g}")
        assert (pure <= LOGMEL_RTOL).mean() >= 0.999, f"{what}: {100 * (pure > LOGMEL_RTOL).mean():.3f} % beyond 1e-4 pure relative"
        assert pure.max() <= LOGMEL_PURE_CAP, f"{what}: max pure relative err {pure.max():.3e}"
    return rel.max(initial=0)


def assert_cep(gpu, ref, what, atol=CEP_ATOL):
    gpu = np.asarray(gpu, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    assert gpu.shape == ref.shape, what
    nan = np.isnan(ref)
    assert np.array_equal(np.isnan(gpu), nan), f"{what}: NaN pattern differs"
    err = np.abs(gpu[~nan] - ref[~nan])
    assert err.max(initial=0) <= atol, f"{what}: max abs err {err.max():.3e}"
    return err.max(initial=0)


# ----------------------------------------------------------------------------------------
# the reference's one real utterance, every stage
# ----------------------------------------------------------------------------------------
@pytest.mark.parametrize("variant", VARIANTS)
def test_ldc93s1_fused(M, golden, variant):
    g = golden("ldc93s1.npz")
    fe = M.MFCC(variant)
    audio = torch.from_numpy(g["audio"]).cuda()
    out = fe(audio).cpu().numpy()
    assert out.shape == (291, 39) and out.dtype == np.float32
    assert_cep(out, g[f"{variant}_out"], f"{variant} out")
    lm = fe(audio, kind="logmel").cpu().numpy()
    assert lm.shape == (291, 27)
    assert_logmel(lm[:, :26], g[f"{variant}_logmel"], f"{variant} ln-mel")
    assert_logmel(lm[:, 26], g[f"{variant}_loge"], f"{variant} ln-energy")
    st = fe(audio, kind="static").cpu().numpy()
    assert_cep(st, g[f"{variant}_static"], f"{variant} static")
    rows = g[f"{variant}_rows"]
    spec = fe(audio, kind="spec").cpu().numpy()
    ref = g[f"{variant}_pow_rows"] if variant == "self5" else g[f"{variant}_mag_rows"]
    scale = np.abs(ref).max(axis=1, keepdims=True)
    assert (np.abs(spec[rows] - ref) <= 2e-6 * scale).all()
    # numpy input -> numpy output, same numbers
    out_np = fe(g["audio"])
    assert isinstance(out_np, np.ndarray) and np.array_equal(out_np, out)


def test_ldc93s1_known_answer(M, golden):
    g = golden("ldc93s1.npz")
    out = M.mfcc39(g["audio"])
    assert np.allclose(out[0, :5], [7.07865742, -25.0219473, -6.32589368, -12.31370909, -4.65412786], atol=1e-3)
    assert abs(float(out.astype(np.float64).sum()) - (-18230.0624550056)) < 0.5


@pytest.mark.parametrize("variant", VARIANTS)
def test_synthetic_lengths(M, golden, variant):
    """Frame-count edge cases (L = 1, 2, 399, 400, 401, 560, 561, 4000, 48000)."""
    g = golden("synthetic.npz")
    fe = M.MFCC(variant)
    for length in g["lengths"]:
        audio = g[f"audio_{length}"]
        out = fe(audio)
        assert out.shape == (O.frame_count(length), 39)
        assert_cep(out, g[f"{variant}_out_{length}"], f"{variant} L={length}")
        if length <= 4000:
            lm = fe(audio, kind="logmel")
            assert_logmel(lm[:, :26], g[f"{variant}_logmel_{length}"], f"{variant} ln-mel L={length}")
            assert_logmel(lm[:, 26], g[f"{variant}_loge_{length}"], f"{variant} ln-E L={length}")


@pytest.mark.parametrize("variant", VARIANTS)
def test_special_inputs(M, golden, variant):
    """Digital silence (ln 0 = -inf -> NaN cepstra, energy floored at eps), silence after
    speech, full-scale alternating samples, pure DC."""
    g = golden("special.npz")
    fe = M.MFCC(variant)
    for name in ("zeros", "tailzeros", "fullscale", "dc"):
        audio = g[f"audio_{name}"]
        out = fe(audio)
        ref = g[f"{variant}_out_{name}"]
        assert_cep(out, ref, f"{variant} {name}")
        lm = fe(audio, kind="logmel")
        rl = g[f"{variant}_logmel_{name}"]
        if name in ("zeros", "tailzeros"):
            assert np.array_equal(np.isneginf(lm[:, :26]), np.isneginf(rl))
            fin = np.isfinite(rl)
            assert np.abs(lm[:, :26][fin] - rl[fin]).max(initial=0) <= 1e-4 * np.maximum(np.abs(rl[fin]), 1).max(initial=1)
        assert_logmel(lm[:, 26], g[f"{variant}_loge_{name}"], f"{variant} ln-E {name}")
    z = fe(g["audio_zeros"])
    assert abs(z[0, 0] - np.log(np.finfo(float).eps)) < 1e-4 and np.isnan(z[:, 1:13]).all()


@pytest.mark.parametrize("variant", VARIANTS)
def test_float32_input(M, golden, variant):
    g = golden("float_input.npz")
    fe = M.MFCC(variant)
    out = fe(g["audio"])
    assert_cep(out, g[f"{variant}_out"], f"{variant} float input")
    lm = fe(g["audio"], kind="logmel")
    assert_logmel(lm[:, :26], g[f"{variant}_logmel"], f"{variant} float ln-mel")


# ----------------------------------------------------------------------------------------
# batches
# ----------------------------------------------------------------------------------------
def test_ragged_batch_equals_single_utterances(M, golden):
    g = golden("synthetic.npz")
    lengths = [int(x) for x in g["lengths"]]
    packed = np.concatenate([g[f"audio_{n}"] for n in lengths])
    for variant in VARIANTS:
        fe = M.MFCC(variant)
        batch = fe(torch.from_numpy(packed).cuda(), lengths=lengths).cpu().numpy()
        singles = np.concatenate([fe(g[f"audio_{n}"]) for n in lengths])
        assert np.array_equal(batch, singles, equal_nan=True)
        ref = np.concatenate([g[f"{variant}_out_{n}"] for n in lengths])
        assert_cep(batch, ref, f"{variant} ragged batch")


def test_long_utterance_tiles_match_oracle(M):
    """An utterance far longer than one workgroup tile: tile seams must be invisible."""
    rng = np.random.default_rng(11)
    audio = np.clip(rng.standard_normal(16000 * 20) * 3000, -32768, 32767).astype(np.int16)
    for variant in VARIANTS:
        out = M.MFCC(variant)(audio)
        ref = O.mfcc39(audio, variant, loop=False)
        assert_cep(out, ref, f"{variant} 20 s utterance")


def test_dense_batch_and_out_argument(M):
    rng = np.random.default_rng(3)
    audio = np.clip(rng.standard_normal((7, 8000)) * 3000, -32768, 32767).astype(np.int16)
    fe = M.MFCC("self5")
    a = torch.from_numpy(audio).cuda()
    out = torch.full((7 * 49, 40), -7.0, device="cuda")
    res = fe(a, out=out)
    assert res.data_ptr() == out.data_ptr()
    torch.cuda.synchronize()
    assert (out[:, 39] == -7.0).all()  # padding column untouched (out_ld = 40)
    ref = np.concatenate([O.mfcc39(audio[i], "self5", loop=False) for i in range(7)])
    assert_cep(out[:, :39].cpu().numpy(), ref, "dense batch")


def test_full_size_batch_properties(M):
    """BASELINE config: 10 000 x 3 s.  Checked through size-independent properties: shape,
    determinism, batch independence (a slice run alone gives the same rows bit for bit),
    finiteness, and the oracle on a sample of utterances."""
    gen = torch.Generator(device="cuda").manual_seed(1234)
    audio = (torch.randn(10000, 48000, generator=gen, device="cuda") * 3000).clamp(-32768, 32767).to(torch.int16)
    fe = M.MFCC("self5")
    out = fe(audio)
    assert out.shape == (10000 * 299, 39)
    out2 = fe(audio)
    assert torch.equal(out, out2)
    assert torch.isfinite(out).all()
    sub = fe(audio[4321:4325].contiguous())
    assert torch.equal(sub, out[4321 * 299:4325 * 299])
    # the tiles left over after whole rounds of the grid (the last 16 of 10 000 on 3 x 256 places) are cut
    # into shorter ones by the layout: same rows as when those utterances run alone, as whole tiles
    assert torch.equal(fe(audio[9980:].contiguous()), out[9980 * 299:])
    for u in list(range(0, 10000, 769)) + [9999]:       # a strided sample of 15 utterances
        ref = O.mfcc39(audio[u].cpu().numpy(), "self5", loop=False)
        assert_cep(out[u * 299:(u + 1) * 299].cpu().numpy(), ref, f"utt {u}")
    # the delta columns of EVERY row are the regression of the static columns of its own utterance
    st = out[:, :13].reshape(10000, 299, 13)
    idx = torch.arange(299, device="cuda")
    def reg(x):
        tap = lambda k: x[:, torch.clamp(idx + k, 0, 298)]
        return (2 * (tap(2) - tap(-2)) + (tap(1) - tap(-1))) / 10
    d = reg(st)
    assert (out[:, 13:26].reshape(10000, 299, 13) - d).abs().max() < 1e-4
    assert (out[:, 26:].reshape(10000, 299, 13) - reg(d)).abs().max() < 1e-4


def test_loadrawdata_round_trip(M, golden, tmp_path):
    """loadrawdata (mfcc-self5.py:32-36): a headerless little-endian int16 file comes back sample
    for sample and feeds the pipeline like the array it was written from."""
    g = golden("ldc93s1.npz")
    path = tmp_path / "utt.raw"
    g["audio"].astype("<i2").tofile(str(path))
    raw = M.loadrawdata(str(path))
    assert raw.dtype == np.int16 and np.array_equal(raw, g["audio"])
    assert_cep(M.mfcc39(raw), g["self5_out"], "loadrawdata -> mfcc39")


# ----------------------------------------------------------------------------------------
# stage by stage, under the reference's own function names
# ----------------------------------------------------------------------------------------
def test_reference_named_stages(M, golden):
    s = golden("stages.npz")
    assert_cep(M.preemphasis(s["preemph_in"]), s["preemph_out"], "preemphasis", 2e-3)
    assert_cep(M.preemphasis(s["preemph_in"], 0.95), s["preemph_out_095"], "preemphasis 0.95", 2e-3)
    assert_cep(M.delta(s["delta_in"]), s["delta_out"], "delta", 1e-5)
    assert_cep(M.delta(s["delta_in"], 3), s["delta_out_pad3"], "delta pad 3", 1e-5)
    assert_cep(M.delta(s["delta_in_1"]), s["delta_out_1"], "delta single frame", 1e-6)
    assert_cep(M.mfcc(s["dct_in"]), s["dct_out12"], "dct 12", 1e-4)
    assert_cep(M.mfcc(list(s["dct_in"]), numcep=13), s["dct_out13"], "dct 13", 1e-4)
    assert_cep(M.dctopt(s["dct_out13"], 22), s["lift_out"], "lifter", 1e-3)
    assert_cep(M.dctopt(s["dct_out13"], 0), s["lift_out_0"], "lifter off", 1e-5)
    assert np.array_equal(M.melbanks(16000, 40, 0, 512), s["melbank_40"])
    assert np.array_equal(M.melbanks(16000, 26, 300, 512), s["melbank_26_low300"])


@pytest.mark.parametrize("variant", VARIANTS)
def test_reference_main_call_chain(M, golden, variant):
    """The body of the reference's main() (mfcc-self5.py:211-229 / mfcc-self15.py:228-246)
    written against this package's names."""
    g = golden("ldc93s1.npz")
    audio = g["audio"]
    fs, frame_time, step_time, num_bins, low_freq, lift_factor = M.fs, M.frame_time, M.step_time, M.num_bins, M.low_freq, M.lift_factor
    audio_pre = M.preemphasis(audio)
    magfft_list, powfft_list, win, num_fft = M.winfft(audio_pre if variant == "self5" else audio, fs, frame_time, step_time)
    melbank = M.melbanks(fs, num_bins, low_freq, num_fft)
    melspec_list, energy_list = M.melspec(powfft_list if variant == "self5" else magfft_list, melbank)
    mfcc_list = M.mfcc(melspec_list, numcep=13)
    mfcclift_list = M.dctopt(mfcc_list, lift_factor)
    cepstrum = M.staticfeatures(energy_list, mfcclift_list)
    delta_features = M.delta(cepstrum)
    delta_delta_features = M.delta(delta_features)
    spectrogram = M.spectrograms(cepstrum, delta_features, delta_delta_features)
    assert num_fft == 512 and np.allclose(win, g["win"]) and len(magfft_list) == 291
    assert sum(1 for _ in powfft_list) == 291 and powfft_list[0].shape == (257,)
    assert_logmel(melspec_list, g[f"{variant}_logmel"], "stagewise ln-mel")
    assert_logmel(energy_list, g[f"{variant}_loge"], "stagewise ln-E")
    assert_cep(spectrogram, g[f"{variant}_out"], "stagewise out")
    # and the fused path agrees with the staged one
    assert_cep(M.MFCC(variant)(audio), spectrogram, "fused vs staged", 2e-3)


def test_spectrum257_and_hamming_extension(M, golden):
    g = golden("ldc93s1.npz")
    audio = g["audio"]
    pw = M.spectrum257(audio)
    ref = O.spectrum257(audio, "self5")
    assert (np.abs(pw - ref) <= 2e-6 * np.abs(ref).max(axis=1, keepdims=True)).all()
    lg = M.spectrum257(audio, log=True)
    assert np.abs(lg - np.log(pw)).max() < 1e-5
    # Hamming-windowed frames (not the reference's behaviour: checked against the oracle's
    # own window, parity with the reference is not claimed for this mode)
    sig = O.preemphasis(audio)
    size, step = 400, 160
    nfr = O.frame_count(len(audio))
    padded = np.zeros((nfr - 1) * step + size)
    padded[:len(sig)] = sig
    idx = np.arange(nfr)[:, None] * step + np.arange(size)[None, :]
    refw = np.abs(np.fft.rfft(padded[idx] * np.hamming(size), 512, axis=1)) ** 2 / 512
    got = M.mfcc39(audio, window="hamming", kind="spec")
    assert (np.abs(got - refw) <= 2e-6 * refw.max(axis=1, keepdims=True)).all()


# ----------------------------------------------------------------------------------------
# context stacking and the .pat container
# ----------------------------------------------------------------------------------------
def test_stacking_bit_exact(M, golden, tmp_path):
    g = golden("stacking.npz")
    for name in ("is1c0001_001", "is1c0001_031"):
        path = tmp_path / f"{name}.pat"
        path.write_bytes(g[f"{name}_pat_bytes"].tobytes())
        data = M.loadPatData(str(path), 12)  # [C, F] like the reference
        assert np.array_equal(data, g[f"{name}_data"])
        labels = g[f"{name}_labels"]
        for framenum in (1, 11, 17):
            ref = g[f"{name}_stack{framenum}"]
            out = M.stack_context(np.ascontiguousarray(data.T), framenum)
            assert out.dtype == np.float32 and np.array_equal(out, ref)
            plab, pdat = M.padUtterance(framenum, labels, data)
            assert pdat.shape == (39, data.shape[1] + framenum - 1)
            lablist, veclist = M.slidingWin(framenum, plab, pdat, -11)
            assert np.array_equal(np.asarray(veclist), ref)
            assert np.array_equal(np.asarray(lablist), g[f"{name}_stacklab{framenum}"])
        out_path = tmp_path / f"{name}_rt.pat"
        M.writePatData(str(out_path), data.T)
        assert out_path.read_bytes() == g[f"{name}_pat_bytes"].tobytes()


def test_stacking_ragged_257x17(M):
    rng = np.random.default_rng(5)
    counts = [1, 5, 40, 299, 8]
    feats = rng.standard_normal((sum(counts), 257)).astype(np.float32)
    out = M.stack_context(torch.from_numpy(feats).cuda(), 17, frame_counts=counts).cpu().numpy()
    ref, r = [], 0
    for n in counts:
        ref.append(O.stack_context(feats[r:r + n], 17))
        r += n
    assert out.shape == (sum(counts), 17 * 257) and np.array_equal(out, np.concatenate(ref))
    with pytest.raises(Exception):
        M.stack_context(feats, 16)


# ----------------------------------------------------------------------------------------
# CMVN
# ----------------------------------------------------------------------------------------
def test_cmvn_sums_and_apply(M):
    rng = np.random.default_rng(8)
    x = (rng.standard_normal((70001, 39)) * 5 + 3).astype(np.float32)
    t = torch.from_numpy(x).cuda()
    sums = M.cmvn_accumulate(t)
    sums = M.cmvn_accumulate(t[:1000], sums)
    n, s, ss = O.cmvn_sums(np.concatenate([x, x[:1000]]))
    got = sums.cpu().numpy()
    assert got[0] == n
    assert np.allclose(got[1:40], s, rtol=1e-12, atol=1e-7) and np.allclose(got[40:], ss, rtol=1e-12)
    mm, vv = M.cmvn_constants(sums)
    rmm, rvv = O.cmvn_constants(n, s, ss)
    assert np.allclose(mm.cpu().numpy(), rmm, rtol=1e-6) and np.allclose(vv.cpu().numpy(), rvv, rtol=1e-6)
    y = M.cmvn_apply(t.clone(), mm, vv).cpu().numpy()
    assert np.abs(y - O.cmvn_apply(x, rmm, rvv)).max() < 1e-5


def test_error_behaviour(M):
    # a valid reference call at another rate runs (general kernel): digital silence at 8 kHz
    z = M.mfcc39(np.zeros(1000, dtype=np.int16), fs=8000)
    assert z.shape == (O.frame_count(1000, 200, 80), 39) and np.isnan(z[:, 1:13]).all() and np.allclose(z[:, 0], np.log(O.EPS64))
    with pytest.raises(Exception, match="FFT lengths"):
        M.mfcc39(np.zeros(1000, dtype=np.int16), fs=400000)
    with pytest.raises(Exception, match="length"):
        M.MFCC("self5")(torch.zeros(10, dtype=torch.int16, device="cuda"), lengths=[10, 0])
    with pytest.raises(ValueError):
        M.MFCC("self5").forward(torch.zeros(10, dtype=torch.int16), None)
    # a row stride beyond the specialised kernel's 32-bit row offsets is refused, not wrapped around
    import ctypes as C
    from mfcc_amd import _capi as capi
    fe = M.MFCC("self5")
    lay = fe.layout([4000])
    a = torch.zeros(4000, dtype=torch.int16, device="cuda")
    o = torch.zeros(64, dtype=torch.float32, device="cuda")
    lib = capi.load()
    rc = lib.mfcc_forward(fe.plan._h, lay._h, C.c_void_p(a.data_ptr()), capi.AUDIO_I16, C.c_void_p(o.data_ptr()), 7_100_000, capi.OUT_MFCC39, None)
    assert rc == capi.ERR_UNSUPPORTED and b"32-bit row offsets" in lib.mfcc_last_error()
    rc = lib.mfcc_forward(fe.plan._h, lay._h, C.c_void_p(a.data_ptr()), capi.AUDIO_I16, C.c_void_p(o.data_ptr()), 38, capi.OUT_MFCC39, None)
    assert rc == capi.ERR_INVALID


# --------------------------------------------------------------------------
# streaming: rows of all pushes + flush == whole-utterance result on the concatenated audio
# --------------------------------------------------------------------------
def _stream_all(s, audio, chunks):
    """Push audio [B, L] in the given chunk sizes (cycled), then flush; returns [B, F, 39]."""
    parts, pos, i = [], 0, 0
    while pos < audio.shape[1]:
        n = min(chunks[i % len(chunks)], audio.shape[1] - pos)
        parts.append(s.push(audio[:, pos:pos + n]))
        pos += n
        i += 1
    parts.append(s.flush())
    return torch.cat(parts, dim=1)


@pytest.mark.gpu
def test_streaming_64_streams_of_10ms_hops():
    """The largest streaming batch of the BASELINE configuration: 64 lock-step streams pushed in
    10 ms hops; every stream equals the offline rows of its own audio and the oracle."""
    import mfcc_amd
    rng = np.random.default_rng(64)
    B, L = 64, 3200 + 57
    audio = np.clip(rng.standard_normal((B, L)) * 3000, -32768, 32767).astype(np.int16)
    s = mfcc_amd.StreamingMFCC("self5", n_streams=B, max_chunk=160)
    got = _stream_all(s, torch.from_numpy(audio).cuda(), [160]).cpu().numpy()
    off = mfcc_amd.MFCC("self5")(torch.from_numpy(audio).cuda()).cpu().numpy().reshape(B, -1, 39)
    np.testing.assert_allclose(got, off, rtol=0, atol=2e-5)
    for b in (0, 17, 63):
        assert_cep(got[b], O.mfcc39(audio[b], "self5", loop=False), f"stream {b} of 64")


@pytest.mark.gpu
@pytest.mark.parametrize("variant", ["self5", "self15"])
@pytest.mark.parametrize("chunks", [[160], [137, 1, 1000, 23], [1600], [4000, 160, 160]])
def test_streaming_equals_offline(variant, chunks):
    import mfcc_amd
    from oracle import mfcc_oracle as O
    rng = np.random.default_rng(7)
    B, L = 3, 5000 + 37
    audio = np.clip(rng.standard_normal((B, L)) * 3000, -32768, 32767).astype(np.int16)
    s = mfcc_amd.StreamingMFCC(variant, n_streams=B, max_chunk=4000)
    got = _stream_all(s, torch.from_numpy(audio).cuda(), chunks).cpu().numpy()
    off = mfcc_amd.MFCC(variant)(torch.from_numpy(audio).cuda()).cpu().numpy().reshape(B, -1, 39)
    assert got.shape == off.shape
    # same kernels for the statics, same regression expression for the deltas
    np.testing.assert_allclose(got, off, rtol=0, atol=2e-5)
    for b in range(B):
        assert_cep(got[b], O.mfcc39(audio[b], variant), f"stream {b}")
    # the object is reusable: a second, shorter utterance after flush
    got2 = _stream_all(s, torch.from_numpy(audio[:, :777]).cuda(), chunks).cpu().numpy()
    for b in range(B):
        assert_cep(got2[b], O.mfcc39(audio[b, :777], variant), f"stream {b} second utterance")


@pytest.mark.gpu
@pytest.mark.parametrize("L", [1, 399, 400, 401, 560, 561, 1040, 1041])
def test_streaming_short_utterances(L):
    import mfcc_amd
    from oracle import mfcc_oracle as O
    rng = np.random.default_rng(L)
    audio = np.clip(rng.standard_normal((2, L)) * 3000, -32768, 32767).astype(np.int16)
    s = mfcc_amd.StreamingMFCC("self5", n_streams=2, max_chunk=320)
    got = _stream_all(s, torch.from_numpy(audio).cuda(), [160, 77]).cpu().numpy()
    for b in range(2):
        ref = O.mfcc39(audio[b], "self5")
        assert got[b].shape == ref.shape
        assert_cep(got[b], ref, f"L={L} stream {b}")


@pytest.mark.gpu
def test_streaming_equals_offline_bit_for_bit():
    """Streamed rows equal the offline rows of the same plan bit for bit (both form delta and
    delta-delta with the same explicitly rounded regression), also with a numcep other than 13 (the
    kernels with a run-time coefficient count)."""
    import mfcc_amd
    rng = np.random.default_rng(12)
    audio = np.clip(rng.standard_normal((3, 7000)) * 3000, -32768, 32767).astype(np.int16)
    dev = torch.from_numpy(audio).cuda()
    for numcep, nb in ((13, 26), (12, 33), (16, 33)):
        s = mfcc_amd.StreamingMFCC("self5", n_streams=3, max_chunk=480, numcep=numcep, num_bins=nb)
        got = _stream_all(s, dev, [160, 480, 33])
        fe = mfcc_amd.MFCC("self5", numcep=numcep, num_bins=nb)
        off = fe(dev)
        assert got.shape == (3, off.shape[0] // 3, 3 * numcep)
        assert torch.equal(got.reshape(-1, 3 * numcep), off)


@pytest.mark.gpu
def test_streaming_float_input_numpy_and_errors():
    import mfcc_amd
    from oracle import mfcc_oracle as O
    rng = np.random.default_rng(3)
    audio = (rng.standard_normal((1, 3000)) * 0.1).astype(np.float32)
    s = mfcc_amd.StreamingMFCC("self5", n_streams=1, max_chunk=800, dtype=torch.float32)
    parts = [s.push(audio[:, i:i + 800]) for i in range(0, 3000, 800)]
    assert all(isinstance(p, np.ndarray) for p in parts)
    got = np.concatenate(parts + [s.flush(as_numpy=True)], axis=1)
    assert_cep(got[0], O.mfcc39(audio[0], "self5"), "float stream")
    with pytest.raises(ValueError):
        s.push(np.zeros((1, 801), np.float32))
    with pytest.raises(ValueError):
        s.push(np.zeros((2, 10), np.float32))
    assert s.flush().shape[1] == 0  # nothing pushed: no rows


# --------------------------------------------------------------------------
# other parameterisations of the same functions (keyword arguments of melbanks / mfcc / dctopt)
# --------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("num_bins,numcep,lift,low", [(40, 13, 22, 0), (20, 13, 0, 100.0), (26, 12, 22, 0), (33, 16, 10, 0)])
def test_other_parameterisations(num_bins, numcep, lift, low):
    import mfcc_amd
    rng = np.random.default_rng(11)
    audio = np.clip(rng.standard_normal(30000) * 3000, -32768, 32767).astype(np.int16)   # 186 frames: the ring of static rows wraps
    st = O.mfcc39(audio, "self5", num_bins=num_bins, low_freq=low, lift_factor=lift, return_stages=True)
    with np.errstate(invalid="ignore"):
        static = O.dctopt(O.dct_ortho(st["logmel"], numcep, loop=False), lift)
        static[:, 0] = st["loge"]
        d = O.delta(static, loop=False)
        ref = np.concatenate([static, d, O.delta(d, loop=False)], axis=1)
    got = mfcc_amd.mfcc39(audio, num_bins=num_bins, numcep=numcep, lift_factor=lift, low_freq=low)
    assert got.shape == ref.shape == (st["out"].shape[0], 3 * numcep)
    assert_cep(got, ref, f"nmel={num_bins} numcep={numcep}")
    lm = mfcc_amd.mfcc39(audio, num_bins=num_bins, numcep=numcep, lift_factor=lift, low_freq=low, kind="logmel")
    assert_logmel(lm[:, :num_bins], st["logmel"], "logmel")


# --------------------------------------------------------------------------
# consumers' data plumbing on the device: CTC batch shaping, .pat pipeline, psf-style call
# --------------------------------------------------------------------------
@pytest.mark.gpu
def test_pad_sequences_and_scalar_norm():
    import mfcc_amd
    rng = np.random.default_rng(5)
    counts = [7, 1, 30, 12]
    seqs = [rng.standard_normal((n, 39)).astype(np.float32) * 5 + 2 for n in counts]
    packed = torch.from_numpy(np.concatenate(seqs)).cuda()
    for maxlen in (None, 10, 40):
        for padding in ("post", "pre"):
            for truncating in ("post", "pre"):
                x, lens = mfcc_amd.pad_sequences(packed, counts, maxlen=maxlen, padding=padding, truncating=truncating, value=-1.5)
                ref, rl = O.pad_sequences(seqs, maxlen, padding, truncating, -1.5)
                assert np.array_equal(lens, rl)
                assert np.array_equal(x.cpu().numpy(), ref), (maxlen, padding, truncating)   # a copy: bit-exact
    with pytest.raises(ValueError):
        mfcc_amd.pad_sequences(packed, counts, padding="middle")
    normed = mfcc_amd.utterance_scalar_norm(packed.clone(), counts).cpu().numpy()
    pos = 0
    for s in seqs:
        np.testing.assert_allclose(normed[pos:pos + len(s)], O.utterance_scalar_norm(s), rtol=0, atol=2e-6)
        pos += len(s)


@pytest.mark.gpu
def test_ctc_feature_batch(golden):
    import mfcc_amd
    rng = np.random.default_rng(9)
    audios = [np.clip(rng.standard_normal(n) * 3000, -32768, 32767).astype(np.int16) for n in (4000, 16000, 7003)]
    x, seq = mfcc_amd.ctc_feature_batch(audios, time_steps=80)
    x = x.cpu().numpy()
    assert x.shape == (3, 80, 39)
    for i, a in enumerate(audios):
        ref = O.utterance_scalar_norm(O.mfcc39(a, "self5"))      # inference_ctc_eri_05.py:291
        n = min(80, ref.shape[0])
        assert seq[i] == ref.shape[0]
        np.testing.assert_allclose(x[i, :n], ref[:n], rtol=0, atol=2e-4)  # 1e-3 cepstral tolerance / std (~10)
        assert np.all(x[i, n:] == 0)


@pytest.mark.gpu
def test_pat_pipeline_preprocess_databuffer(golden, tmp_path):
    from mfcc_amd import datasets as D
    import mfcc_amd
    g = golden("stacking.npz")
    root = str(tmp_path) + "/"
    names = ["is1c0001_001.pat", "is1c0001_031.pat"]
    feats = [np.asarray(g[n[:-4] + "_data"]).T.astype(np.float32) for n in names]   # golden data is [C, F]
    counts = [f.shape[0] for f in feats]
    D.write_pat_corpus(root, names, torch.from_numpy(np.concatenate(feats)).cuda(), counts)
    import os
    os.makedirs(root + D.label_dir, exist_ok=True)
    for n in names:
        np.savetxt(root + D.label_dir + n[:-4] + ".txt", np.asarray(g[n[:-4] + "_labels"]))
        with open(root + D.data_dir + n, "rb") as fh:                     # the writer reproduces the reference's file
            assert fh.read() == bytes(np.asarray(g[n[:-4] + "_pat_bytes"]).tobytes())
    old = mfcc_amd.frontend.coeffs
    try:
        labels, vecs = D.Preprocess(17, 1, names, root, -11)
        assert np.array_equal(np.asarray(vecs), np.asarray(g["is1c0001_031_stack17"]))
        assert np.array_equal(np.asarray(labels), np.asarray(g["is1c0001_031_stacklab17"]))
        lb, db = D.Databuffer(2, 11, names, root, -11)
        assert np.array_equal(db, np.concatenate([g["is1c0001_001_stack11"], g["is1c0001_031_stack11"]]))
        assert np.array_equal(np.asarray(lb), np.concatenate([g["is1c0001_001_stacklab11"], g["is1c0001_031_stacklab11"]]))
        mb = D.createMinibatchData(16, 11, db)
        assert mb.shape == (16, 11 * 39) and mb.dtype == np.float64 and np.array_equal(mb, db[:16].astype(np.float64))
    finally:
        mfcc_amd.frontend.coeffs = old


@pytest.mark.gpu
def test_psf_style_call_matches_documented_algorithm():
    from mfcc_amd import psf
    rng = np.random.default_rng(2)
    audio = np.clip(rng.standard_normal(8000) * 3000, -32768, 32767).astype(np.int16)
    audio[3000:3600] = 0                                   # a stretch of digital silence: exact-zero mel bands
    feat = psf.mfcc(audio, samplerate=16000, winlen=0.025, winstep=0.01, numcep=13, nfilt=26, nfft=512, lowfreq=0,
                    highfreq=None, preemph=0.97, ceplifter=22, appendEnergy=True)
    st = O.mfcc39(audio, "self5", return_stages=True)
    mel = np.exp(st["logmel"])
    lm = np.log(np.where(mel == 0, O.EPS64, mel))          # the floor the package applies before the log
    ref = O.dctopt(O.dct_ortho(lm, 13, loop=False), 22)
    ref[:, 0] = st["loge"]
    assert np.isfinite(feat).all()
    assert_cep(feat, ref, "psf.mfcc")
    d = psf.delta(feat, 2)
    np.testing.assert_allclose(d, O.delta(feat.astype(np.float64), loop=False), rtol=0, atol=1e-4)
    full = psf.get_wav_mfcc(audio, 16000)
    assert full.shape == (feat.shape[0], 39) and np.allclose(full[:, :13], feat, atol=1e-6)
    # the other arguments of that call signature (general kernel): upper band edge, FFT length,
    # window function, the DCT's own c0, regression width
    _mag, pw, _w, _n = O.winfft(O.preemphasis(audio), loop=False, num_fft=1024)
    win = np.hamming(400)
    idx = np.arange(pw.shape[0])[:, None] * 160 + np.arange(400)[None, :]
    padded = np.zeros(idx.max() + 1)
    padded[:len(audio)] = O.preemphasis(audio)
    pw = np.square(np.abs(np.fft.rfft(padded[idx] * win, 1024, axis=1))) / 1024
    mel = pw @ O.melbanks(16000, 20, 100, 1024, highfreq=4000).T
    lm = np.log(np.where(mel == 0, O.EPS64, mel))
    ref = O.dctopt(O.dct_ortho(lm, 12, loop=False), 22)      # appendEnergy=False: c0 stays the DCT's
    feat = psf.mfcc(audio, nfft=1024, nfilt=20, numcep=12, lowfreq=100, highfreq=4000, appendEnergy=False, winfunc=np.hamming)
    assert_cep(feat, ref, "psf.mfcc(nfft=1024, highfreq=4000, appendEnergy=False, winfunc=np.hamming)")
    np.testing.assert_allclose(psf.delta(feat, 3), O.delta(feat.astype(np.float64), 3, loop=False), rtol=0, atol=1e-4)
    with pytest.raises(ValueError):
        psf.mfcc(audio, winfunc=np.blackman)
    with pytest.raises(Exception):
        psf.mfcc(audio, highfreq=9000)


@pytest.mark.gpu
def test_examples_run_like_the_reference_main(golden, tmp_path, monkeypatch, capsys):
    """examples/ hold the counterparts of the reference's call sites; run them in-process."""
    import runpy
    import scipy.io.wavfile as wav
    g = golden("ldc93s1.npz")
    path = str(tmp_path / "utt.wav")
    wav.write(path, 16000, np.asarray(g["audio"], dtype=np.int16))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for variant in ("self5", "self15"):
        for extra in ([], ["--fused"]):
            monkeypatch.setattr(sys, "argv", ["mfcc_self5_main.py", path, "--variant", variant, "--pat", str(tmp_path / "o.pat")] + extra)
            runpy.run_path(os.path.join(root, "examples", "mfcc_self5_main.py"), run_name="__main__")
            out = np.load(str(tmp_path / f"utt.{variant}.mfcc39.npy"))
            assert_cep(out, g[f"{variant}_out"], f"example {variant} {extra}")
    monkeypatch.setattr(sys, "argv", ["stream_chunks.py", path, "--streams", "2"])
    runpy.run_path(os.path.join(root, "examples", "stream_chunks.py"), run_name="__main__")
    assert "max |stream - offline|" in capsys.readouterr().out
    monkeypatch.setattr(sys, "argv", ["ctc_batch.py", path, path, "--time-steps", "300"])
    runpy.run_path(os.path.join(root, "examples", "ctc_batch.py"), run_name="__main__")
    assert "(2, 300, 39)" in capsys.readouterr().out


@pytest.mark.gpu
@pytest.mark.parametrize("variant", VARIANTS)
def test_8khz_geometry_with_512_point_fft(variant):
    """25 ms / 10 ms at 8 kHz = 200/80-sample frames, FFT kept at 512 points (the CTC scripts' call,
    inference_ctc_eri_05.py:96,253-257).  Checked against the restatement with num_fft=512."""
    import mfcc_amd
    rng = np.random.default_rng(21)
    lengths = [1, 199, 200, 201, 281, 8000, 12345]
    audios = [np.clip(rng.standard_normal(n) * 3000, -32768, 32767).astype(np.int16) for n in lengths]
    fe = mfcc_amd.MFCC(variant, fs=8000, num_fft=512)
    assert (fe.plan.frame_size, fe.plan.frame_step, fe.plan.num_fft) == (200, 80, 512)
    out = fe(np.concatenate(audios), lengths=lengths)
    pos = 0
    for a in audios:
        ref = O.mfcc39(a, variant, fs=8000, num_fft=512)
        assert_cep(out[pos:pos + ref.shape[0]], ref, f"8 kHz L={len(a)}")
        pos += ref.shape[0]
    assert pos == out.shape[0]
    # streaming at 8 kHz
    s = mfcc_amd.StreamingMFCC(variant, n_streams=1, max_chunk=400, fs=8000, num_fft=512)
    got = _stream_all(s, torch.from_numpy(audios[-1][None, :]).cuda(), [80, 333]).cpu().numpy()[0]
    assert_cep(got, O.mfcc39(audios[-1], variant, fs=8000, num_fft=512), "8 kHz stream")
    # the reference's own winfft takes a 256-point FFT at this frame size (mfcc-self5.py:79-84):
    # that configuration is checked against the reference's goldens in tests/test_gpu_general.py
    assert mfcc_amd.MFCC(variant, fs=8000).plan.num_fft == 256


@pytest.mark.gpu
def test_matlab_spec_anal_mode():
    """spec_anal (anal_recog_nn_liste.m:497-532): parity with the reference itself is unpinned (no
    Matlab); held to the numpy/scipy restatement of the cited lines."""
    from mfcc_amd import matlab as M
    rng = np.random.default_rng(17)
    lengths = [5000, 399, 400, 240, 16000, 561]
    audios = [np.clip(rng.standard_normal(n) * 3000, -32768, 32767).astype(np.int16) for n in lengths]
    got = M.spec_anal(np.concatenate(audios), lengths=lengths)
    pos = 0
    for a in audios:
        ref = O.spec_anal(a, M.initdft["bpre"])
        n = ref.shape[0]
        assert n == max(0, (len(a) - 240) // 160)
        assert_logmel(got[pos:pos + n], ref, f"spec_anal L={len(a)}")
        pos += n
    assert pos == got.shape[0]
    # the filter alone, including utterance boundaries, and without pre-emphasis
    y = M.fir_centered(torch.from_numpy(np.concatenate(audios)).cuda(), lengths, M.initdft["bpre"], 7).cpu().numpy()
    pos = 0
    for a in audios:
        full = np.convolve(a.astype(np.float64), np.asarray(M.initdft["bpre"]))[7:7 + len(a)]
        np.testing.assert_allclose(y[pos:pos + len(a)], full, rtol=0, atol=2e-2)   # float32 sums of 15 terms ~1e4
        pos += len(a)
    ref = O.spec_anal(audios[0], M.initdft["bpre"], use_preemp=False)
    assert_logmel(M.spec_anal(audios[0], init=dict(use_preemp=False)), ref, "spec_anal without pre-emphasis")
    mm, vv = -ref.mean(0), 1.0 / ref.std(0)
    normed = M.spec_anal(audios[0], init=dict(use_preemp=False), mm=mm, vv=vv)
    np.testing.assert_allclose(normed, (ref + mm) * vv, rtol=0, atol=2e-3)


@pytest.mark.gpu
@pytest.mark.parametrize("variant,dtype", [("self5", np.int16), ("self15", np.int16), ("self5", np.float32)])
def test_randomised_ragged_batch_around_tile_boundaries(variant, dtype):
    """Seeded ragged batch whose frame counts straddle the fused kernel's tile size (300 output
    frames, 4-frame halos) and the batch size (16), next to very short utterances."""
    import mfcc_amd
    rng = np.random.default_rng(1234)
    frames = [1, 2, 4, 5, 15, 16, 17, 31, 32, 33] + list(range(290, 314, 3)) + [599, 600, 601, 607, 608, 609, 905]
    lengths = [400 + 160 * (f - 1) - int(rng.integers(0, 160)) if f > 1 else int(rng.integers(1, 401)) for f in frames]
    lengths += [int(v) for v in rng.integers(1, 3000, size=40)]
    audios = []
    for n in lengths:
        a = rng.standard_normal(n) * 3000
        audios.append(np.clip(a, -32768, 32767).astype(np.int16) if dtype == np.int16 else (a / 32768).astype(np.float32))
    order = rng.permutation(len(lengths))
    audios = [audios[i] for i in order]
    lengths = [lengths[i] for i in order]
    out = mfcc_amd.MFCC(variant)(np.concatenate(audios), lengths=lengths)
    pos = 0
    for a in audios:
        ref = O.mfcc39(a, variant, loop=False)
        assert_cep(out[pos:pos + ref.shape[0]], ref, f"L={len(a)}")
        pos += ref.shape[0]
    assert pos == out.shape[0]


@pytest.mark.gpu
def test_forward_is_graph_capturable():
    """mfcc_forward only enqueues one kernel on the caller's stream (no allocation, no host
    synchronisation), so a launch-bound caller can capture it in a HIP graph and replay it."""
    import mfcc_amd
    rng = np.random.default_rng(4)
    lengths = [3000, 16000, 777]
    audio = torch.from_numpy(np.clip(rng.standard_normal(sum(lengths)) * 3000, -32768, 32767).astype(np.int16)).cuda()
    fe = mfcc_amd.MFCC("self5")
    lay = fe.layout(lengths)
    out = torch.zeros((lay.total_frames, 39), dtype=torch.float32, device="cuda")
    eager = fe.forward(audio, lay).clone()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fe.forward(audio, lay, out=out)          # warm-up on the side stream
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fe.forward(audio, lay, out=out)
    out.zero_()
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, eager)
    audio2 = torch.from_numpy(np.clip(rng.standard_normal(sum(lengths)) * 3000, -32768, 32767).astype(np.int16)).cuda()
    audio.copy_(audio2)                          # new samples in the captured buffer, same launch
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, fe.forward(audio2, lay))
    # the one-launch stacked spectrum (more LDS than a kernel gets by default: granted at plan creation) captures too
    ctx = torch.zeros((lay.total_frames, 17 * 257), dtype=torch.float32, device="cuda")
    eager = mfcc_amd.spectrum_stack(fe, audio, lay, 17).clone()
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2):
        mfcc_amd.spectrum_stack(fe, audio, lay, 17, out=ctx)
    ctx.zero_()
    g2.replay()
    torch.cuda.synchronize()
    assert torch.equal(ctx, eager)


@pytest.mark.gpu
def test_streaming_large_chunks_span_several_tiles():
    """A push that brings more frames than one tile of the fused kernel holds (here 625 and 312)."""
    import mfcc_amd
    rng = np.random.default_rng(8)
    audio = np.clip(rng.standard_normal((2, 150000)) * 3000, -32768, 32767).astype(np.int16)
    s = mfcc_amd.StreamingMFCC("self5", n_streams=2, max_chunk=100000)
    got = _stream_all(s, torch.from_numpy(audio).cuda(), [100000, 50000]).cpu().numpy()
    off = mfcc_amd.MFCC("self5")(torch.from_numpy(audio).cuda()).cpu().numpy().reshape(2, -1, 39)
    assert got.shape == off.shape
    np.testing.assert_allclose(got, off, rtol=0, atol=2e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.int16, torch.float32])
def test_views_into_larger_buffers_ignore_what_lies_around_them(dtype):
    """The staging loads fetch aligned 16-byte chunks, clamped (a tile's first batch) or range-checked by a buffer
    descriptor (the later ones), so they touch bytes in front of and behind the samples they were given.  The
    same samples at element offsets 0, 1, 3, 5 and 8 of a larger tensor whose other elements are NaN (float32) or
    full-scale noise (int16) must give the same bits as a tensor of their own: nothing outside the view counts,
    whatever 16-byte phase and sample parity the view starts on."""
    import mfcc_amd
    rng = np.random.default_rng(77)
    lengths = [1, 399, 401, 2959, 48000, 52345, 7, 16001, 97000]
    n = sum(lengths)
    host = rng.standard_normal(n) * 3000
    if dtype == torch.int16:
        own = torch.from_numpy(np.clip(host, -32768, 32767).astype(np.int16)).cuda()
    else:
        own = torch.from_numpy((host / 32768).astype(np.float32)).cuda()
    fe = mfcc_amd.MFCC("self5")
    lay = fe.layout(lengths)
    ref = fe.forward(own, lay, kind="mfcc39").clone()
    assert torch.isfinite(ref).all()
    for off in (0, 1, 3, 5, 8):
        big = torch.empty(n + 4096 + off, dtype=dtype, device="cuda")
        if dtype == torch.int16:
            big[:] = torch.from_numpy(rng.choice(np.array([-32768, 32767], dtype=np.int16), size=big.numel())).cuda()
        else:
            big[:] = float("nan")
        big[off:off + n] = own
        out = fe.forward(big[off:off + n], lay, kind="mfcc39")
        assert torch.equal(out, ref), f"{dtype}, view at element {off}: rows differ from those of a tensor of its own"
