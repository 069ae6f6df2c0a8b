"""GPU parity of the general-geometry kernel: every (fs, frame_time, step_time) the reference's
winfft accepts (mfcc-self5.py:75-84), any pad_fac of delta (:169), other filterbank / cepstrum
sizes.  Checked against golden vectors the reference's own functions produced at 8 kHz
(200/80 samples, 256-point FFT), 44.1 kHz (1102/441, 2048-point), 20 ms frames (320/160, 512-point)
and 32 ms / 8 ms (512/128, 512-point), and against the CPU oracle.  Tolerances as in
test_gpu_parity.py."""
import numpy as np
import pytest
import torch

from oracle import mfcc_oracle as O
from test_gpu_parity import VARIANTS, assert_cep, assert_logmel, _stream_all

pytestmark = pytest.mark.gpu

GEOMS = ("8k", "44k", "16k_20ms", "16k_32ms_8ms")


@pytest.fixture(scope="module")
def M():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    import mfcc_amd
    return mfcc_amd


def _synth(seed, n):
    rng = np.random.default_rng(seed)
    return np.clip(rng.standard_normal(n) * 3000, -32768, 32767).astype(np.int16)


@pytest.mark.parametrize("gname", GEOMS)
@pytest.mark.parametrize("variant", VARIANTS)
def test_other_frame_geometries_against_reference_goldens(M, golden, gname, variant):
    """One ragged batch per geometry holding every golden utterance; each utterance's rows against
    what the reference's own pipeline produced for it."""
    g = golden("geometry.npz")
    fs, frame_time, step_time = (float(v) for v in g[f"{gname}_cfg"])
    lengths = [int(v) for v in g[f"{gname}_lengths"]]
    audios = [g[f"{gname}_audio_{n}"] for n in lengths]
    fe = M.MFCC(variant, fs=fs, frame_time=frame_time, step_time=step_time)
    size, step, nfft = O.frame_geometry(fs, frame_time, step_time)
    assert (fe.plan.frame_size, fe.plan.frame_step, fe.plan.num_fft) == (size, step, nfft) and nfft == int(g[f"{gname}_nfft"])
    # 8 kHz: 200/80-sample frames are a framing of the specialised kernel, whose 512-point transform
    # holds the 256-point spectrum on its even bins; the other framings run the general kernel
    # (and 320/160-sample frames, frame_time = 20 at 16 kHz, are a third framing of the specialised kernel)
    assert fe.plan.general == (gname not in ("8k", "16k_20ms"))
    packed = torch.from_numpy(np.concatenate(audios)).cuda()
    lay = fe.layout(lengths)
    out = fe.forward(packed, lay).cpu().numpy()
    lm = fe.forward(packed, lay, kind="logmel").cpu().numpy()
    st = fe.forward(packed, lay, kind="static").cpu().numpy()
    spec = fe.forward(packed, lay, kind="spec").cpu().numpy()
    assert spec.shape[1] == nfft // 2 + 1
    for u, n in enumerate(lengths):
        a, b = lay.frame_offsets[u], lay.frame_offsets[u + 1]
        ref = g[f"{gname}_{variant}_out_{n}"]
        assert b - a == ref.shape[0] == O.frame_count(n, size, step)
        assert_cep(out[a:b], ref, f"{gname} {variant} L={n}")
        assert_cep(st[a:b], ref[:, :13], f"{gname} {variant} static L={n}")
        assert_logmel(lm[a:b, :26], g[f"{gname}_{variant}_logmel_{n}"], f"{gname} {variant} ln-mel L={n}")
        assert_logmel(lm[a:b, 26], g[f"{gname}_{variant}_loge_{n}"], f"{gname} {variant} ln-E L={n}")
        key = f"{gname}_{variant}_spec_{n}"
        if key in g.files:
            sref = g[key]
            assert (np.abs(spec[a:b] - sref) <= 2e-6 * np.abs(sref).max(axis=1, keepdims=True)).all()
    # float32 samples take the same path
    outf = fe(packed.to(torch.float32), lengths=lengths).cpu().numpy()
    assert_cep(outf, out, f"{gname} float input", atol=2e-3 if variant == "self5" else 1e-3)
    # and the general kernel agrees with whichever kernel ran (for 8 kHz: the specialised one)
    gen = M.MFCC(variant, fs=fs, frame_time=frame_time, step_time=step_time, kernel="general")
    assert gen.plan.general
    assert_cep(gen.forward(packed, gen.layout(lengths)).cpu().numpy(), out, f"{gname} general vs selected kernel", atol=2e-3)
    lspec = fe.forward(packed, lay, kind="logspec").cpu().numpy()
    fin = spec > 1e-3 * spec.max(axis=1, keepdims=True)
    assert np.abs(lspec[fin] - np.log(spec[fin])).max() < 1e-5


@pytest.mark.parametrize("variant", VARIANTS)
def test_general_kernel_at_the_default_framing(M, golden, variant):
    """kernel='general' at 400/160/512: the reference's own utterance and the frame-count edge
    cases through the general kernel, against the goldens and against the specialised kernel."""
    g = golden("ldc93s1.npz")
    fe = M.MFCC(variant, kernel="general")
    fast = M.MFCC(variant)
    assert fe.plan.general and not fast.plan.general
    audio = torch.from_numpy(g["audio"]).cuda()
    out = fe(audio).cpu().numpy()
    assert_cep(out, g[f"{variant}_out"], f"general {variant} out")
    assert_cep(out, fast(audio).cpu().numpy(), "general vs specialised", atol=2e-3)
    lm = fe(audio, kind="logmel").cpu().numpy()
    assert_logmel(lm[:, :26], g[f"{variant}_logmel"], f"general {variant} ln-mel")
    assert_logmel(lm[:, 26], g[f"{variant}_loge"], f"general {variant} ln-energy")
    s = golden("synthetic.npz")
    lengths = [int(v) for v in s["lengths"]]
    got = fe(np.concatenate([s[f"audio_{n}"] for n in lengths]), lengths=lengths)
    pos = 0
    for n in lengths:
        ref = s[f"{variant}_out_{n}"]
        assert_cep(got[pos:pos + ref.shape[0]], ref, f"general {variant} L={n}")
        pos += ref.shape[0]
    assert pos == got.shape[0]
    # digital silence: -inf ln-mel, NaN cepstra, floored energy -- the reference's pattern
    sp = golden("special.npz")
    for name in ("zeros", "tailzeros", "dc", "fullscale"):
        assert_cep(fe(sp[f"audio_{name}"]), sp[f"{variant}_out_{name}"], f"general {variant} {name}")
        lmz = fe(sp[f"audio_{name}"], kind="logmel")
        assert_logmel(lmz[:, :26], sp[f"{variant}_logmel_{name}"], f"general {variant} {name} ln-mel")


@pytest.mark.parametrize("pad_fac", [1, 3, 5])
def test_fused_delta_with_other_pad_fac(M, golden, pad_fac):
    """delta(features, pad_fac) is general in the reference (mfcc-self5.py:169-179; the pad_fac=3
    golden of the stage test pins the oracle): whole rows [c | d | dd] with that regression width,
    over a ragged batch with utterances shorter than the window and longer than a tile."""
    s = golden("stages.npz")
    np.testing.assert_allclose(O.delta(s["delta_in"], 3), s["delta_out_pad3"], rtol=0, atol=1e-12)
    lengths = [1, 400, 561, 1300, 48000, 120000]
    audios = [_synth(50 + i, n) for i, n in enumerate(lengths)]
    fe = M.MFCC("self5", pad_fac=pad_fac)
    assert fe.plan.general
    got = fe(np.concatenate(audios), lengths=lengths)
    pos = 0
    for a in audios:
        static = O.mfcc39(a, "self5", loop=False, return_stages=True)["static"]
        d = O.delta(static, pad_fac, loop=False)
        ref = O.spectrograms(static, d, O.delta(d, pad_fac, loop=False))
        assert_cep(got[pos:pos + ref.shape[0]], ref, f"pad_fac={pad_fac} L={len(a)}")
        pos += ref.shape[0]
    assert pos == got.shape[0]


def test_other_filterbank_and_cepstrum_sizes(M):
    """64 filters / 20 cepstra / low_freq 100 (outside the specialised kernel's 40 / 16): against
    the oracle's stage functions."""
    a = _synth(77, 20000)
    fe = M.MFCC("self5", num_bins=64, numcep=20, low_freq=100, lift_factor=0)
    assert fe.plan.general
    got = fe(a)
    _mag, pw, _w, nfft = O.winfft(O.preemphasis(a), loop=False)
    logmel, loge = O.melspec(pw, O.melbanks(16000, 64, 100, nfft), loop=False)
    static = O.staticfeatures(loge, O.dctopt(O.dct_ortho(logmel, 20, loop=False), 0))
    d = O.delta(static, loop=False)
    assert_cep(got, O.spectrograms(static, d, O.delta(d, loop=False)), "64 filters, 20 cepstra")
    assert_logmel(fe(a, kind="logmel")[:, :64], logmel, "64 filters ln-mel")


def test_explicit_fft_length_shorter_than_the_frame(M):
    """np.fft.rfft(chunk, n) with n < len(chunk) uses the first n samples (the call at
    mfcc-self5.py:96 with another num_fft): 400-sample frames, 256-point FFT."""
    a = _synth(78, 5000)
    fe = M.MFCC("self15", num_fft=256)
    assert fe.plan.general and fe.plan.num_fft == 256
    assert_cep(fe(a), O.mfcc39(a, "self15", num_fft=256, loop=False), "num_fft=256 < frame")


@pytest.mark.parametrize("variant", VARIANTS)
def test_reference_named_winfft_at_other_rates(M, golden, variant):
    """winfft(audio, fs, frame_time, step_time) keeps the reference's signature for any rate: the
    returned spectra, window and num_fft at 8 kHz, 44.1 kHz and 20 ms frames."""
    g = golden("geometry.npz")
    for gname in ("8k", "44k", "16k_20ms"):
        fs, frame_time, step_time = (float(v) for v in g[f"{gname}_cfg"])
        n = int(g[f"{gname}_lengths"][1])
        audio = g[f"{gname}_audio_{n}"]
        src = O.preemphasis(audio).astype(np.float32) if variant == "self5" else audio
        mag, pw, win, nfft = M.winfft(src, fs, frame_time, step_time)
        assert nfft == int(g[f"{gname}_nfft"]) and win.shape == (O.frame_geometry(fs, frame_time, step_time)[0],)
        sref = g[f"{gname}_{variant}_spec_{n}"]
        got = pw if variant == "self5" else mag
        assert (np.abs(got - sref) <= 4e-6 * np.abs(sref).max(axis=1, keepdims=True)).all()


def test_streaming_at_8khz_with_the_derived_256_point_fft(M):
    """StreamingMFCC at the framing winfft derives for 8 kHz (200/80 samples, 256-point FFT), on the
    specialised and on the general kernel: pushes + flush equal the offline rows."""
    a = _synth(79, 12345)
    for variant, kernel in [(v, k) for v in VARIANTS for k in ("auto", "general")]:
        s = M.StreamingMFCC(variant, n_streams=2, max_chunk=400, fs=8000, kernel=kernel)
        assert s.plan.general == (kernel == "general") and s.plan.num_fft == 256
        both = torch.from_numpy(np.stack([a, a[::-1].copy()])).cuda()
        got = _stream_all(s, both, [80, 333, 7]).cpu().numpy()
        assert_cep(got[0], O.mfcc39(a, variant, fs=8000, loop=False), f"8 kHz stream {variant}")
        assert_cep(got[1], O.mfcc39(a[::-1], variant, fs=8000, loop=False), f"8 kHz stream {variant} (2)")


def test_layouts_and_plans_of_different_framings_do_not_mix(M):
    """A layout's tiles are cut for its plan's framing: another framing is refused, not run."""
    a = torch.from_numpy(_synth(80, 30000)).cuda()
    fe16, fe8 = M.MFCC("self5"), M.MFCC("self5", fs=8000)
    with pytest.raises(M.frontend.capi.MfccError):
        fe8.forward(a, fe16.layout([30000]))
    with pytest.raises(M.frontend.capi.MfccError):
        fe16.forward(a, fe8.layout([30000]))
    # the one-launch stacked spectrum is built for 512-point plans; others take the two-kernel route inside spectrum_stack
    lay8 = fe8.layout([30000])
    two = M.stack_context(fe8.forward(a, lay8, kind="logspec"), 5, frame_counts=np.diff(lay8.frame_offsets))
    assert two.shape[1] == 5 * 129 and torch.equal(M.spectrum_stack(fe8, a, lay8, 5), two)
    # same framing, other kernel: the general kernel's short tiles suit the specialised kernel
    gen = M.MFCC("self5", kernel="general")
    assert_cep(fe16.forward(a, gen.layout([30000])).cpu().numpy(), fe16(a).cpu().numpy(), "general-cut layout", atol=0)
    # ... but not the other way round when a tile is longer than the general kernel holds
    with pytest.raises(M.frontend.capi.MfccError):
        gen.forward(a, fe16.layout([30000]))
    # configurations no kernel covers still say so
    with pytest.raises(M.frontend.capi.MfccError):
        M.MFCC("self5", fs=192000)            # 4800-sample frames: 8192-point FFT
    with pytest.raises(M.frontend.capi.MfccError):
        M.MFCC("self5", num_fft=500)          # not a power of two
    with pytest.raises(M.frontend.capi.MfccError):
        M.MFCC("self5", fs=30)                # int(30 * 0.010) = 0 samples per step


def test_largest_tables_leave_no_room_for_delta_rows(M):
    """2048-point FFT, 128 filters, 64 cepstra, pad_fac 9: the FFT buffers and filter rows leave too little of a
    workgroup's LDS for the static and delta rows of even a four-frame tile with its 36 halo rows, so
    the plan has no fused MFCC-39 (mfcc_plan_supports says so, mfcc_forward refuses it with a clear
    error) but serves every other kind; the Python front-end builds MFCC-39 from the static rows and
    mfcc_delta per utterance."""
    import ctypes as C
    from mfcc_amd import _capi as capi
    a, b = _synth(91, 30000), _synth(92, 9000)
    fe = M.MFCC("self5", fs=44100, frame_time=40, step_time=10, num_bins=128, numcep=64, pad_fac=9)
    assert fe.plan.general and fe.plan.num_fft == 2048
    lib = capi.load()
    assert lib.mfcc_plan_supports(fe.plan._h, capi.OUT_MFCC39) == 0
    assert lib.mfcc_plan_supports(fe.plan._h, capi.OUT_STATIC) == 1 and lib.mfcc_plan_supports(fe.plan._h, capi.OUT_SPEC) == 1
    packed = torch.from_numpy(np.concatenate([a, b])).cuda()
    lay = fe.layout([len(a), len(b)])
    out = torch.zeros((lay.total_frames, 192), dtype=torch.float32, device="cuda")
    rc = lib.mfcc_forward(fe.plan._h, lay._h, C.c_void_p(packed.data_ptr()), capi.AUDIO_I16, C.c_void_p(out.data_ptr()), 192,
                          capi.OUT_MFCC39, None)
    assert rc == capi.ERR_UNSUPPORTED and b"mfcc_delta" in lib.mfcc_last_error()
    got = fe(packed, lengths=[len(a), len(b)]).cpu().numpy()
    pos = 0
    for x in (a, b):
        _mag, pw, _w, nfft = O.winfft(O.preemphasis(x), fs=44100, frame_time=40, step_time=10, loop=False)
        logmel, loge = O.melspec(pw, O.melbanks(44100, 128, 0, nfft), loop=False)
        static = O.staticfeatures(loge, O.dctopt(O.dct_ortho(logmel, 64, loop=False), 22))
        d = O.delta(static, 9, loop=False)
        ref = O.spectrograms(static, d, O.delta(d, 9, loop=False))
        assert_cep(got[pos:pos + ref.shape[0]], ref, "2048-point, 128 filters, 64 cepstra, pad_fac 9")
        pos += ref.shape[0]
    assert pos == got.shape[0]
    # the same tables with the default regression window fuse, and so does a long window on narrow rows
    for kw in (dict(numcep=64), dict(numcep=13, pad_fac=14)):
        fe2 = M.MFCC("self5", fs=44100, frame_time=40, step_time=10, num_bins=128, **kw)
        assert lib.mfcc_plan_supports(fe2.plan._h, capi.OUT_MFCC39) == 1
        assert torch.isfinite(fe2(packed, lengths=[len(a), len(b)])).all()


@pytest.mark.parametrize("fs,nfft", [(22050, 1024), (32000, 1024), (48000, 2048), (44100, 2048)])
@pytest.mark.parametrize("variant", VARIANTS)
def test_1024_and_2048_point_plans_against_the_oracle(M, fs, nfft, variant):
    """The frames of 22.05 / 32 kHz (551 / 800 samples, 1024-point FFT: 32 lanes per frame, two frames per wave) and of
    44.1 / 48 kHz (1102 / 1200 samples, 2048 points: 64 lanes per frame) on a ragged batch with frame-count edge cases
    (one sample, one frame, one sample more than a frame, several tiles): every kind against the oracle's stages."""
    size, step, n = O.frame_geometry(fs, 25, 10)
    assert n == nfft
    lengths = [1, size - 1, size, size + 1, size + step, 9 * step + size + 7, 150 * step + 11, 333 * step]
    audios = [_synth(500 + i, L) for i, L in enumerate(lengths)]
    fe = M.MFCC(variant, fs=fs)
    assert fe.plan.general and fe.plan.num_fft == nfft
    packed = torch.from_numpy(np.concatenate(audios)).cuda()
    lay = fe.layout(lengths)
    out = fe.forward(packed, lay).cpu().numpy()
    lm = fe.forward(packed, lay, kind="logmel").cpu().numpy()
    st = fe.forward(packed, lay, kind="static").cpu().numpy()
    spec = fe.forward(packed, lay, kind="spec").cpu().numpy()
    assert spec.shape[1] == nfft // 2 + 1
    for u, a in enumerate(audios):
        r0, r1 = lay.frame_offsets[u], lay.frame_offsets[u + 1]
        ref = O.mfcc39(a, variant, fs=fs, return_stages=True)
        assert r1 - r0 == ref["out"].shape[0] == O.frame_count(len(a), size, step)
        assert_cep(out[r0:r1], ref["out"], f"fs={fs} {variant} L={len(a)}")
        assert_cep(st[r0:r1], ref["out"][:, :13], f"fs={fs} {variant} static L={len(a)}")
        assert_logmel(lm[r0:r1, :26], ref["logmel"], f"fs={fs} {variant} ln-mel L={len(a)}")
        assert_logmel(lm[r0:r1, 26], ref["loge"], f"fs={fs} {variant} ln-E L={len(a)}")
        sref = ref["pow"] if variant == "self5" else ref["mag"]
        assert (np.abs(spec[r0:r1] - sref) <= 2e-6 * np.abs(sref).max(axis=1, keepdims=True)).all()
    outf = fe(packed.to(torch.float32), lengths=lengths).cpu().numpy()
    assert_cep(outf, out, f"fs={fs} float input", atol=2e-3 if variant == "self5" else 1e-3)
    # Hamming window (the extension) and another pad_fac / filterbank on the same kernel, against the oracle's pieces
    a = audios[-2]
    few = M.MFCC(variant, fs=fs, window="hamming", num_bins=40, numcep=20, pad_fac=3)
    got = few(a)
    pre = O.preemphasis(a) if variant == "self5" else a.astype(np.float64)
    nf = O.frame_count(len(a), size, step)
    padded = np.concatenate([pre, np.zeros((nf - 1) * step + size - len(a))])
    frames = padded[np.arange(nf)[:, None] * step + np.arange(size)[None, :]] * np.hamming(size)
    mag = np.abs(np.fft.rfft(frames, nfft, axis=1))
    pw = mag ** 2 / nfft
    logmel, loge = O.melspec(pw if variant == "self5" else mag, O.melbanks(fs, 40, 0, nfft), loop=False)
    static = O.staticfeatures(loge, O.dctopt(O.dct_ortho(logmel, 20, loop=False), 22))
    d = O.delta(static, 3, loop=False)
    assert_cep(got, O.spectrograms(static, d, O.delta(d, 3, loop=False)), f"fs={fs} hamming, 40 filters, 20 cepstra, pad_fac 3")


def _oracle_rows(a, variant, fs, frame_time, step_time, num_bins, numcep, pad_fac, lift):
    """MFCC rows of one utterance from the oracle's stage functions with free sizes (mfcc-self5.py:211-229 with
    numcep, num_bins and pad_fac as arguments instead of the script's constants)."""
    cfg = O.VARIANTS[variant]
    sig = O.preemphasis(a, cfg["preemph"]) if cfg["preemph"] is not None else np.asarray(a)
    mag, pw, _w, nfft = O.winfft(sig, fs, frame_time, step_time, loop=False)
    spec = pw if cfg["spectrum"] == "power" else mag
    logmel, loge = O.melspec(spec, O.melbanks(fs, num_bins, 0, nfft), loop=False)
    with np.errstate(invalid="ignore", divide="ignore"):
        static = O.staticfeatures(loge, O.dctopt(O.dct_ortho(logmel, numcep, loop=False), lift))
        d = O.delta(static, pad_fac, loop=False)
        dd = O.delta(d, pad_fac, loop=False)
    return spec, logmel, loge, O.spectrograms(static, d, dd)


def test_random_configurations_against_the_oracle(M):
    """A seeded sweep over what winfft, melbanks, mfcc and delta accept: sample rates 4 - 48 kHz, frame and step
    times that give FFT lengths from 64 to 2048 points (all three kernels of the general path and, where the draw
    hits 400/160 or 200/80, the specialised one), 8 - 64 filters, 2 - 24 cepstra, pad_fac 1 - 6, both variants, int16
    and float32 samples, ragged batches with one-frame utterances: spectrum, ln-mel, ln E and MFCC rows of every
    utterance against the oracle's stages."""
    rng = np.random.default_rng(20240611)
    seen = set()
    for trial in range(28):
        fs = float(rng.choice([4000, 8000, 11025, 16000, 22050, 32000, 44100, 48000]))
        frame_time = float(rng.choice([8, 10, 16, 20, 25, 32, 40]))
        step_time = float(rng.choice([4, 5, 8, 10, 12.5, 16]))
        size, step, nfft = O.frame_geometry(fs, frame_time, step_time)
        if size > 2048 or step > size or step < 1 or nfft < 64:
            continue
        num_bins = int(rng.integers(8, min(64, nfft // 8) + 1))
        numcep = int(rng.integers(2, min(24, num_bins) + 1))
        pad_fac = int(rng.integers(1, 7))
        variant = VARIANTS[trial % 2]
        lift = float(rng.choice([0, 22]))
        lengths = [1, size, size + 1, int(rng.integers(size, 40 * size)), int(rng.integers(size, 80 * size)), 3 * step + size - 1]
        audios = [_synth(9000 + 10 * trial + i, L) for i, L in enumerate(lengths)]
        fe = M.MFCC(variant, fs=fs, frame_time=frame_time, step_time=step_time, num_bins=num_bins, numcep=numcep, pad_fac=pad_fac,
                    lift_factor=lift)
        assert (fe.plan.frame_size, fe.plan.frame_step, fe.plan.num_fft) == (size, step, nfft)
        seen.add((nfft, bool(fe.plan.general)))
        what = f"trial {trial}: fs={fs:g} {frame_time:g}/{step_time:g} ms -> {size}/{step}/{nfft}, {num_bins} filters, {numcep} cep, pad_fac {pad_fac}, {variant}"
        packed = np.concatenate(audios)
        as_float = trial % 3 == 2
        t = torch.from_numpy(packed.astype(np.float32) if as_float else packed).cuda()
        lay = fe.layout(lengths)
        out = fe.forward(t, lay).cpu().numpy()
        lm = fe.forward(t, lay, kind="logmel").cpu().numpy()
        spec = fe.forward(t, lay, kind="spec").cpu().numpy()
        for u, a in enumerate(audios):
            r0, r1 = lay.frame_offsets[u], lay.frame_offsets[u + 1]
            src = a.astype(np.float32) if as_float else a
            sref, lmref, leref, oref = _oracle_rows(src, variant, fs, frame_time, step_time, num_bins, numcep, pad_fac, lift)
            assert r1 - r0 == oref.shape[0] == O.frame_count(len(a), size, step), what
            assert (np.abs(spec[r0:r1] - sref) <= 3e-6 * np.abs(sref).max(axis=1, keepdims=True) + 1e-30).all(), what
            assert_logmel(lm[r0:r1, :num_bins], lmref, what + f" ln-mel L={len(a)}")
            assert_logmel(lm[r0:r1, num_bins], leref, what + f" ln-E L={len(a)}")
            # float32 samples: the reference would pre-emphasise in float32 too; cepstra to 2e-3 as in the other tests
            assert_cep(out[r0:r1], oref, what + f" L={len(a)}", atol=2e-3 if as_float else CEP_ATOL_SWEEP)
    # the draw reached every kernel of the general path
    assert {n for n, g in seen if g} >= {1024, 2048} and any(n <= 512 for n, g in seen if g), sorted(seen)


CEP_ATOL_SWEEP = 1e-3


def test_20ms_frames_on_the_specialised_kernel(M):
    """frame_time = 20 at 16 kHz (320/160 samples, 512-point FFT) runs the specialised kernel for the kinds of
    mfcc_forward: against the oracle on a ragged batch, with the statistics entry, float32 samples and a run-time numcep;
    stacked rows are refused there (mfcc_forward + mfcc_stack_context serve them) and streams take the three-launch form."""
    rng = np.random.default_rng(77)
    lengths = [1, 319, 320, 321, 480, 5000, 48000, 160 * 301 + 320, 100000] + [int(v) for v in rng.integers(400, 70000, size=40)]
    audios = [_synth(700 + i, L) for i, L in enumerate(lengths)]
    packed = torch.from_numpy(np.concatenate(audios)).cuda()
    for variant in VARIANTS:
        fe = M.MFCC(variant, frame_time=20)
        assert not fe.plan.general and (fe.plan.frame_size, fe.plan.frame_step, fe.plan.num_fft) == (320, 160, 512)
        lay = fe.layout(lengths)
        out = fe.forward(packed, lay).cpu().numpy()
        lm = fe.forward(packed, lay, kind="logmel").cpu().numpy()
        spec = fe.forward(packed, lay, kind="spec").cpu().numpy()
        st = fe.forward(packed, lay, kind="static").cpu().numpy()
        for u in range(0, len(lengths), 3):
            r0, r1 = lay.frame_offsets[u], lay.frame_offsets[u + 1]
            ref = O.mfcc39(audios[u], variant, frame_time=20, return_stages=True)
            assert r1 - r0 == ref["out"].shape[0]
            assert_cep(out[r0:r1], ref["out"], f"20 ms {variant} L={lengths[u]}")
            assert_cep(st[r0:r1], ref["out"][:, :13], f"20 ms {variant} static L={lengths[u]}")
            assert_logmel(lm[r0:r1, :26], ref["logmel"], f"20 ms {variant} ln-mel L={lengths[u]}")
            assert_logmel(lm[r0:r1, 26], ref["loge"], f"20 ms {variant} ln-E L={lengths[u]}")
            sref = ref["pow"] if variant == "self5" else ref["mag"]
            assert (np.abs(spec[r0:r1] - sref) <= 2e-6 * np.abs(sref).max(axis=1, keepdims=True)).all()
        gen = M.MFCC(variant, frame_time=20, kernel="general")
        assert gen.plan.general
        assert_cep(gen.forward(packed, gen.layout(lengths)).cpu().numpy(), out, f"20 ms {variant} general vs specialised", atol=2e-3)
        sums = torch.zeros(79, dtype=torch.float64, device="cuda")
        assert torch.equal(fe.forward_sums(packed, lay, sums), torch.from_numpy(out).cuda())
        want = M.cmvn_accumulate(torch.from_numpy(out).cuda()).cpu().numpy()
        assert sums[0].item() == want[0] and np.allclose(sums.cpu().numpy()[1:], want[1:], rtol=1e-4, atol=1e-2)
        outf = fe(packed.to(torch.float32), lengths=lengths).cpu().numpy()
        assert_cep(outf, out, f"20 ms {variant} float input", atol=2e-3 if variant == "self5" else 1e-3)
    fe12 = M.MFCC("self5", frame_time=20, numcep=12)
    o12 = np.asarray(fe12(audios[6]))
    assert o12.shape[1] == 36 and not fe12.plan.general
    assert_cep(o12[:, :12], O.mfcc39(audios[6], "self5", frame_time=20)[:, :12], "20 ms numcep 12")
    fe = M.MFCC("self5", frame_time=20)
    with pytest.raises(RuntimeError):
        M.spectrum_stack(fe, packed, fe.layout(lengths), 9)
    # streaming: pushes of uneven size, then the flush, equal the offline rows of the same audio
    a = audios[7]
    s = M.StreamingMFCC("self5", n_streams=2, max_chunk=4000, frame_time=20)
    assert not s.plan.general
    both = torch.from_numpy(np.stack([a, a[::-1].copy()])).cuda()
    got = _stream_all(s, both, [1000, 37, 4000, 160, 3999]).cpu().numpy()
    np.testing.assert_allclose(got[0], np.asarray(fe(a)), rtol=0, atol=2e-5)
    np.testing.assert_allclose(got[1], np.asarray(fe(a[::-1].copy())), rtol=0, atol=2e-5)
