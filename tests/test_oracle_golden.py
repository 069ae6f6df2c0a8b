"""The CPU oracle (oracle/mfcc_oracle.py) against the golden vectors that
oracle/gen_golden.py produced by running the reference's own functions."""
import numpy as np
import pytest

from oracle import mfcc_oracle as O

VARIANTS = ("self5", "self15")


def close(a, b, tol=1e-12):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape
    assert np.array_equal(np.isnan(a), np.isnan(b))
    assert np.array_equal(np.isinf(a), np.isinf(b))
    ok = np.isfinite(b)
    scale = np.maximum(1.0, np.abs(b[ok]))
    assert np.all(np.abs(a[ok] - b[ok]) <= tol * scale), np.max(np.abs(a[ok] - b[ok]) / scale)
    inf = np.isinf(b)
    assert np.array_equal(np.sign(a[inf]), np.sign(b[inf]))


@pytest.mark.parametrize("variant", VARIANTS)
@pytest.mark.parametrize("loop", (True, False))
def test_ldc93s1_all_stages(golden, variant, loop):
    g = golden("ldc93s1.npz")
    st = O.mfcc39(g["audio"], variant, loop=loop, return_stages=True)
    rows = g[f"{variant}_rows"]
    tol = 0.0 if loop else 1e-12
    for k in ("mag", "pow", "cep", "lifted"):
        close(st[k][rows], g[f"{variant}_{k}_rows"], max(tol, 1e-13))
    for k in ("logmel", "loge", "static", "delta", "delta2", "out"):
        close(st[k], g[f"{variant}_{k}"], max(tol, 1e-13))
    # the reference computes the pre-emphasised signal in both variants (self15:228) but
    # self15 does not use it (self15:230)
    close(O.preemphasis(g["audio"])[:512], g[f"{variant}_signal_head"], 0.0)
    if variant == "self15":
        assert np.array_equal(st["signal"], g["audio"].astype(np.float64))
    close(st["melbank"], g["melbank"], 0.0)
    assert st["out"].shape == (291, 39)


def test_ldc93s1_known_answers(golden):
    """Sanity values recorded when the reference was first run."""
    g = golden("ldc93s1.npz")
    o5 = O.mfcc39(g["audio"], "self5")
    assert np.allclose(o5[0, :5], [7.07865742, -25.0219473, -6.32589368, -12.31370909, -4.65412786], atol=1e-7)
    assert abs(o5.sum() - (-18230.0624550056)) < 1e-6
    o15 = O.mfcc39(g["audio"], "self15")
    assert np.allclose(o15[0, :5], [9.3522746, -3.23988571, 2.32075978, -1.79079103, 1.90944701], atol=1e-7)
    assert abs(o15.sum() - (-7228.4603629374)) < 1e-6


def test_loop_form_is_bit_exact_on_ldc93s1(golden):
    g = golden("ldc93s1.npz")
    for variant in VARIANTS:
        out = O.mfcc39(g["audio"], variant, loop=True)
        assert np.array_equal(out, g[f"{variant}_out"])


@pytest.mark.parametrize("variant", VARIANTS)
def test_synthetic_lengths(golden, variant):
    g = golden("synthetic.npz")
    for length in g["lengths"]:
        audio = g[f"audio_{length}"]
        ref = g[f"{variant}_out_{length}"]
        assert ref.shape[0] == O.frame_count(length)
        out = O.mfcc39(audio, variant, loop=True)
        close(out, ref, 1e-13)
        close(O.mfcc39(audio, variant, loop=False), ref, 1e-11)
        if length <= 4000:
            st = O.mfcc39(audio, variant, return_stages=True)
            close(st["logmel"], g[f"{variant}_logmel_{length}"], 1e-13)
            close(st["loge"], g[f"{variant}_loge_{length}"], 1e-13)
        if length in (401, 561):
            st = O.mfcc39(audio, variant, return_stages=True)
            close(st["pow"], g[f"{variant}_pow_{length}"], 1e-13)
            close(st["mag"], g[f"{variant}_mag_{length}"], 1e-13)


def test_frame_count_formula():
    for length, frames in [(1, 1), (2, 1), (399, 1), (400, 1), (401, 2), (560, 2), (561, 3), (4000, 24), (48000, 299), (46797, 291)]:
        assert O.frame_count(length) == frames


@pytest.mark.parametrize("variant", VARIANTS)
def test_special_inputs(golden, variant):
    g = golden("special.npz")
    for name in ("zeros", "tailzeros", "fullscale", "dc"):
        st = O.mfcc39(g[f"audio_{name}"], variant, return_stages=True)
        close(st["out"], g[f"{variant}_out_{name}"], 1e-11)
        close(st["logmel"], g[f"{variant}_logmel_{name}"], 1e-11)
        close(st["loge"], g[f"{variant}_loge_{name}"], 1e-13)
    # digital silence: ln(eps) energy, everything else NaN
    z = g["self5_out_zeros"]
    assert abs(z[0, 0] - np.log(np.finfo(float).eps)) < 1e-12 and np.isnan(z[0, 1:13]).all()


@pytest.mark.parametrize("variant", VARIANTS)
def test_float32_input(golden, variant):
    g = golden("float_input.npz")
    st = O.mfcc39(g["audio"], variant, return_stages=True)
    close(st["out"], g[f"{variant}_out"], 1e-12)
    close(st["logmel"], g[f"{variant}_logmel"], 1e-12)


def test_stage_functions(golden):
    g = golden("stages.npz")
    close(O.delta(g["delta_in"]), g["delta_out"], 0.0)
    close(O.delta(g["delta_in"], 3), g["delta_out_pad3"], 0.0)
    close(O.delta(g["delta_in"], loop=False), g["delta_out"], 1e-14)
    close(O.delta(g["delta_in_1"]), g["delta_out_1"], 0.0)
    close(O.dct_ortho(g["dct_in"], 12), g["dct_out12"], 0.0)
    close(O.dct_ortho(g["dct_in"], 13), g["dct_out13"], 0.0)
    close(g["dct_in"] @ O.dct_matrix(26, 13), g["dct_out13"], 1e-13)
    close(O.dctopt(g["dct_out13"], 22), g["lift_out"], 0.0)
    close(O.dctopt(g["dct_out13"], 0), g["lift_out_0"], 0.0)
    close(O.melbanks(16000, 40, 0, 512), g["melbank_40"], 0.0)
    close(O.melbanks(16000, 26, 300, 512), g["melbank_26_low300"], 0.0)
    close(O.melbanks(8000, 26, 0, 256), g["melbank_8k_26_256"], 0.0)
    close(O.preemphasis(g["preemph_in"]), g["preemph_out"], 0.0)
    close(O.preemphasis(g["preemph_in"], 0.95), g["preemph_out_095"], 0.0)


def test_melbank_structure():
    edges = O.mel_bin_edges()
    assert edges.astype(int).tolist() == [0, 2, 4, 7, 10, 13, 16, 20, 24, 29, 34, 40, 46, 53, 60, 68, 77, 87, 97,
                                          109, 122, 136, 152, 169, 188, 209, 231, 256]
    bank = O.melbanks()
    assert np.count_nonzero(bank) == 459
    assert not bank[:, 0].any() and not bank[:, 256].any()
    assert (np.count_nonzero(bank, axis=0) <= 2).all()


def test_pat_container_and_stacking(golden):
    g = golden("stacking.npz")
    for name in ("is1c0001_001", "is1c0001_031"):
        feats, hdr = O.read_pat(g[f"{name}_pat_bytes"].tobytes(), 39)
        assert hdr[1:] == (100000, 156, 838)
        assert np.array_equal(feats.T, g[f"{name}_data"])
        assert O.write_pat(feats) == g[f"{name}_pat_bytes"].tobytes()
        for framenum in (1, 11, 17):
            ref = g[f"{name}_stack{framenum}"]
            out = O.stack_context(feats, framenum)
            assert out.shape == ref.shape == (feats.shape[0], framenum * 39)
            assert np.array_equal(out, ref)
            labels = g[f"{name}_labels"]
            assert np.array_equal(g[f"{name}_stacklab{framenum}"], labels[: feats.shape[0]])


def test_cmvn_helpers():
    rng = np.random.default_rng(0)
    x = rng.standard_normal((500, 39)) * 3 + 1
    n, s, ss = O.cmvn_sums(x)
    mm, vv = O.cmvn_constants(n, s, ss)
    y = O.cmvn_apply(x, mm, vv)
    assert np.allclose(y.mean(0), 0, atol=1e-12) and np.allclose(y.std(0), 1, atol=1e-12)


def test_cnn_tiles_and_shuffled_order(golden):
    """The oracle's tile-form windows against the reference's CNN slidingWin / createMinibatchData
    (CNN_DenseNet40_6.py:86-131), plain and shuffled."""
    g = golden("stacking_cnn.npz")
    feats = np.ascontiguousarray(g["data"].T)            # the reference carries [C, F]
    labels = g["labels"]
    if labels.shape[0] != feats.shape[0]:
        labels = labels[:-1]
    for framenum in (11, 17):
        for randomint, tag in ((-11, "m11"), (3, "3")):
            order = None if randomint == -11 else O.random_order(randomint, feats.shape[0])
            tiles = O.stack_context_tiles(feats, framenum, order)
            ref = g[f"tiles_n{framenum}_r{tag}"]
            assert tiles.dtype == np.float32 and np.array_equal(tiles, ref)
            want_lab = labels if order is None else labels[order]
            assert np.array_equal(want_lab, g[f"labels_n{framenum}_r{tag}"])
            mb = g[f"minibatch_n{framenum}_r{tag}"]
            assert mb.shape == (16, 39, framenum, 1) and np.array_equal(mb[..., 0], ref[:16].astype(np.float64))
            # the two layouts hold the same windows
            fm = O.stack_context(feats, framenum)
            sel = fm if order is None else fm[order]
            assert np.array_equal(sel.reshape(-1, framenum, 39).transpose(0, 2, 1), ref)


GEOMS = ("8k", "44k", "16k_20ms", "16k_32ms_8ms")


@pytest.mark.parametrize("gname", GEOMS)
@pytest.mark.parametrize("variant", VARIANTS)
def test_other_frame_geometries(golden, gname, variant):
    """winfft derives frame size, step and FFT length from (fs, frame_time, step_time)
    (self5:75-84): 8 kHz -> 200/80/256, 44.1 kHz -> 1102/441/2048, 20 ms -> 320/160/512."""
    g = golden("geometry.npz")
    fs, frame_time, step_time = (float(v) for v in g[f"{gname}_cfg"])
    size, step, nfft = O.frame_geometry(fs, frame_time, step_time)
    assert nfft == int(g[f"{gname}_nfft"])
    for length in g[f"{gname}_lengths"]:
        audio = g[f"{gname}_audio_{length}"]
        st = O.mfcc39(audio, variant, fs=fs, frame_time=frame_time, step_time=step_time, loop=False, return_stages=True)
        assert st["out"].shape == (O.frame_count(length, size, step), 39)
        close(st["out"], g[f"{gname}_{variant}_out_{length}"], 1e-10)
        close(st["logmel"], g[f"{gname}_{variant}_logmel_{length}"], 1e-11)
        close(st["loge"], g[f"{gname}_{variant}_loge_{length}"], 1e-12)
        key = f"{gname}_{variant}_spec_{length}"
        if key in g.files:
            close(st["pow"] if variant == "self5" else st["mag"], g[key], 1e-12)
