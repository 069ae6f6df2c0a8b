"""GPU parity tests of the pieces around the fused kernel: the CNN scripts' window layout and
shuffled order (bit-exact against goldens from the reference's own CNN slidingWin /
createMinibatchData), the batched .pat buffers, bit-reproducible CMVN sums, device guards."""
import os

import numpy as np
import pytest
import torch

from oracle import mfcc_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def M():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    import mfcc_amd
    return mfcc_amd


def _tag(randomint):
    return "m11" if randomint == -11 else str(randomint)


def test_cnn_tiles_bit_exact(M, golden):
    """CNN_DenseNet40_6.py:86-131 through the same-named functions of mfcc_amd.cnn."""
    from mfcc_amd import cnn
    g = golden("stacking_cnn.npz")
    data, labels = g["data"], g["labels"]                    # [C, F] like the reference
    for framenum in (11, 17):
        for randomint in (-11, 3):
            plab, pdat = cnn.padUtterance(framenum, labels, data)
            lablist, tiles = cnn.slidingWin(framenum, plab, pdat, randomint)
            ref = g[f"tiles_n{framenum}_r{_tag(randomint)}"]
            tiles = np.asarray(tiles)
            assert tiles.dtype == np.float32 and tiles.shape == ref.shape == (data.shape[1], 39, framenum)
            assert np.array_equal(tiles, ref)
            assert np.array_equal(np.asarray(lablist), g[f"labels_n{framenum}_r{_tag(randomint)}"])
            mb = cnn.createMinibatchData(16, framenum, tiles)
            assert mb.dtype == np.float64 and np.array_equal(mb, g[f"minibatch_n{framenum}_r{_tag(randomint)}"])
    # iterating the datalist gives one [C, N] slice per window, as the reference's list does
    assert next(iter(tiles)).shape == (39, 17)


def test_stack_layouts_orders_and_ragged(M):
    rng = np.random.default_rng(12)
    counts = [1, 0, 5, 40, 299, 8]
    n = sum(counts)
    for ncol in (39, 257, 4):
        feats = rng.standard_normal((n, ncol)).astype(np.float32)
        dev = torch.from_numpy(feats).cuda()
        ref_fm, ref_tile, r = [], [], 0
        for c in counts:
            if c:
                ref_fm.append(O.stack_context(feats[r:r + c], 17))
                ref_tile.append(O.stack_context_tiles(feats[r:r + c], 17))
            r += c
        ref_fm, ref_tile = np.concatenate(ref_fm), np.concatenate(ref_tile)
        assert np.array_equal(M.stack_context(dev, 17, frame_counts=counts).cpu().numpy(), ref_fm)
        tile = M.stack_context(dev, 17, frame_counts=counts, layout="tile").cpu().numpy()
        assert tile.shape == (n, ncol, 17) and np.array_equal(tile, ref_tile)
        order = rng.permutation(n)[: n // 2]
        assert np.array_equal(M.stack_context(dev, 17, frame_counts=counts, layout="tile", order=order).cpu().numpy(), ref_tile[order])
        assert np.array_equal(M.stack_context(dev, 17, frame_counts=counts, order=torch.from_numpy(order).cuda()).cpu().numpy(), ref_fm[order])
    with pytest.raises(ValueError):
        M.stack_context(dev, 17, frame_counts=counts, order=[n])
    with pytest.raises(ValueError):
        M.stack_context(dev, 17, frame_counts=counts, layout="rows")


def test_batched_databuffers_match_per_utterance(M, golden, tmp_path):
    """Databuffer (MLP_5hidlayer_40.py:163-172, CNN_DenseNet40_6.py:162-171) as one launch over the
    packed corpus equals the per-utterance Preprocess calls, and the goldens."""
    from mfcc_amd import cnn, datasets as D
    g = golden("stacking.npz")
    gc = golden("stacking_cnn.npz")
    root = str(tmp_path) + "/"
    names = ["is1c0001_001.pat", "is1c0001_031.pat"]
    os.makedirs(root + D.data_dir, exist_ok=True)
    os.makedirs(root + D.label_dir, exist_ok=True)
    for n in names:
        with open(root + D.data_dir + n, "wb") as fh:
            fh.write(np.asarray(g[n[:-4] + "_pat_bytes"]).tobytes())
        np.savetxt(root + D.label_dir + n[:-4] + ".txt", np.asarray(g[n[:-4] + "_labels"]))
    lb, db = D.Databuffer(2, 17, names, root, -11)
    assert np.array_equal(db, np.concatenate([g["is1c0001_001_stack17"], g["is1c0001_031_stack17"]]))
    assert np.array_equal(np.asarray(lb), np.concatenate([g["is1c0001_001_stacklab17"], g["is1c0001_031_stacklab17"]]))
    for randomint in (-11, 3):
        lb, tiles = cnn.Databuffer(2, 11, names, root, randomint)
        per = [cnn.Preprocess(11, i, names, root, randomint) for i in range(2)]
        assert np.array_equal(tiles, np.concatenate([np.asarray(p[1]) for p in per]))
        assert np.array_equal(np.asarray(lb), np.concatenate([np.asarray(p[0]) for p in per]))
        assert np.array_equal(tiles[:gc[f"tiles_n11_r{_tag(randomint)}"].shape[0]], gc[f"tiles_n11_r{_tag(randomint)}"])


def test_cmvn_sums_are_bit_reproducible_and_match_float64(M):
    """Fixed reduction order: two accumulations of the same rows give the same bits; the values
    equal a float64 numpy reduction to rounding."""
    rng = np.random.default_rng(8)
    for n, d in ((70001, 39), (1000, 257), (3, 1), (600000, 39)):
        x = (rng.standard_normal((n, d)) * 5 + 3).astype(np.float32)
        t = torch.from_numpy(x).cuda()
        a = M.cmvn_accumulate(t)
        b = M.cmvn_accumulate(t)
        assert torch.equal(a, b)
        cnt, s, ss = O.cmvn_sums(x)
        got = a.cpu().numpy()
        assert got[0] == cnt
        assert np.allclose(got[1:1 + d], s, rtol=1e-12, atol=1e-6) and np.allclose(got[1 + d:], ss, rtol=1e-12)
    # accumulation over chunks adds up
    c = M.cmvn_accumulate(t[:1000])
    c = M.cmvn_accumulate(t[1000:], c)
    assert np.allclose(c.cpu().numpy(), got, rtol=1e-13)


def test_cmvn_contiguous_and_strided_rows_agree(M):
    """Rows back to back take the 16-byte-piece kernels, a column slice of a wider array (or a base
    that is not 16-byte aligned) the row-wise ones: same sums to float64 rounding, same normalised
    values to the bit; element counts that are not multiples of four, widths that share factors
    with four."""
    rng = np.random.default_rng(33)
    for n, d in ((70001, 39), (5003, 257), (999, 40), (4097, 6), (3, 39), (2, 1), (100003, 13)):
        x = (rng.standard_normal((n, d)) * 4 - 2).astype(np.float32)
        wide = torch.zeros((n, d + 3), dtype=torch.float32, device="cuda")
        wide[:, :d] = torch.from_numpy(x).cuda()
        flat = wide[:, :d].contiguous()
        shifted = torch.empty(n * d + 1, dtype=torch.float32, device="cuda")[1:].view(n, d)   # base 4 bytes off a 16-byte boundary
        shifted.copy_(flat)
        cnt, s, ss = O.cmvn_sums(x)
        for t in (flat, wide[:, :d], shifted):
            got = M.cmvn_accumulate(t).cpu().numpy()
            assert got[0] == cnt
            assert np.allclose(got[1:1 + d], s, rtol=1e-12, atol=1e-6) and np.allclose(got[1 + d:], ss, rtol=1e-12)
        sums = M.cmvn_accumulate(flat)
        assert torch.equal(sums, M.cmvn_accumulate(flat, static=True))
        mm, vv = M.cmvn_constants(sums)
        a = M.cmvn_apply(flat.clone(), mm, vv)
        b = M.cmvn_apply(wide.clone()[:, :d], mm, vv)
        c = M.cmvn_apply(shifted, mm, vv)
        assert torch.equal(a, b) and torch.equal(a, c)
        ref = (x.astype(np.float32) + mm.cpu().numpy()) * vv.cpu().numpy()
        assert np.array_equal(a.cpu().numpy(), ref.astype(np.float32))


def test_cmvn_chain_on_mfcc_output(M):
    """forward -> accumulate -> constants -> apply on the product's own MFCC rows, checked against
    float64 numpy statistics of the oracle's features (a strided sample of the utterances)."""
    rng = np.random.default_rng(21)
    lengths = [int(v) for v in rng.integers(2000, 20000, size=64)]
    audios = [np.clip(rng.standard_normal(n) * 3000, -32768, 32767).astype(np.int16) for n in lengths]
    fe = M.MFCC("self5")
    lay = fe.layout(lengths)
    feats = fe.forward(torch.from_numpy(np.concatenate(audios)).cuda(), lay)
    sums = M.cmvn_allreduce(M.cmvn_accumulate(feats))          # world size 1: identity
    mm, vv = M.cmvn_constants(sums)
    raw = feats.cpu().numpy().astype(np.float64)
    normed = M.cmvn_apply(feats, mm, vv).cpu().numpy()
    assert np.abs(normed.mean(0)).max() < 1e-4 and np.abs(normed.std(0) - 1).max() < 1e-4
    cnt, s, ss = O.cmvn_sums(raw)
    rmm, rvv = O.cmvn_constants(cnt, s, ss)
    fo = lay.frame_offsets
    for u in range(0, 64, 9):
        ref = O.cmvn_apply(O.mfcc39(audios[u], "self5", loop=False), rmm, rvv)
        np.testing.assert_allclose(normed[fo[u]:fo[u + 1]], ref, rtol=0, atol=1e-3)   # 1e-3 cepstral tolerance / std (>= 1)


def test_frame_layout_accepts_zero_counts_and_rejects_forward(M):
    from mfcc_amd import sequences
    counts = [3, 0, 2]
    x = torch.arange(5 * 4, dtype=torch.float32, device="cuda").reshape(5, 4)
    pad, lens = sequences.pad_sequences(x, counts, maxlen=3)
    assert pad.shape == (3, 3, 4) and (pad[1] == 0).all() and np.array_equal(lens, counts)
    lay = M.frontend.frames_layout("cuda", counts)
    with pytest.raises(M.frontend.capi.MfccError):
        M.MFCC("self5").forward(torch.zeros(0, dtype=torch.int16, device="cuda"), lay)


def test_device_guards(M):
    fe = M.MFCC("self5", device="cuda")                      # 'cuda' = the current device
    assert fe.plan.device.index == torch.cuda.current_device()
    with pytest.raises(ValueError):
        M.MFCC("self5", device="cpu")
    if torch.cuda.device_count() < 2:
        return
    rng = np.random.default_rng(1)
    audio = np.clip(rng.standard_normal(5000) * 3000, -32768, 32767).astype(np.int16)
    fe1 = M.MFCC("self5", device="cuda:1")
    out1 = fe1(torch.from_numpy(audio).to("cuda:1"))          # current device stays 0
    assert out1.device.index == 1 and torch.cuda.current_device() == 0
    assert np.array_equal(out1.cpu().numpy(), M.MFCC("self5", device="cuda:0")(audio))
    with pytest.raises(ValueError):
        fe1.forward(torch.from_numpy(audio).to("cuda:0"), fe1.layout([5000]))


def test_loadrawdata_round_trip(M, tmp_path):
    rng = np.random.default_rng(2)
    audio = np.clip(rng.standard_normal(1234) * 3000, -32768, 32767).astype(np.int16)
    path = str(tmp_path / "utt.raw")
    audio.astype("<i2").tofile(path)
    back = M.loadrawdata(path)
    assert back.dtype == np.int16 and np.array_equal(back, audio)


def test_streaming_64_streams(M):
    rng = np.random.default_rng(64)
    B, L = 64, 2400 + 11
    audio = np.clip(rng.standard_normal((B, L)) * 3000, -32768, 32767).astype(np.int16)
    s = M.StreamingMFCC("self5", n_streams=B, max_chunk=160)
    dev = torch.from_numpy(audio).cuda()
    parts = [s.push(dev[:, i:i + 160]) for i in range(0, L, 160)] + [s.flush()]
    got = torch.cat(parts, dim=1).cpu().numpy()
    for b in (0, 17, 63):
        ref = O.mfcc39(audio[b], "self5", loop=False)
        assert got[b].shape == ref.shape and np.abs(got[b] - ref).max() <= 1e-3
    off = M.MFCC("self5")(dev).cpu().numpy().reshape(B, -1, 39)
    np.testing.assert_allclose(got, off, rtol=0, atol=2e-5)


def test_log_spectrum_against_oracle(M, golden):
    g = golden("ldc93s1.npz")
    for variant in ("self5", "self15"):
        lg = M.spectrum257(g["audio"], log=True, **M.VARIANTS[variant])
        ref = O.spectrum257(g["audio"], variant, log=True)
        fin = np.isfinite(ref)
        assert np.array_equal(np.isfinite(lg), fin)
        # ln of a float32 spectrum value: the spectrum is good to 2e-6 of the row maximum, so bins
        # far below the row maximum carry that absolute error into their logarithm
        spec = np.exp(ref)
        tol = 2e-6 * spec.max(axis=1, keepdims=True) / np.maximum(spec, 1e-300) + 2e-6 * np.maximum(np.abs(ref), 1)
        assert (np.abs(lg - ref)[fin] <= tol[fin]).all()


@pytest.mark.parametrize("variant,log", [("self5", True), ("self15", False)])
def test_spectrum_stack_one_launch_equals_two_kernels(M, variant, log):
    """mfcc_forward_stacked (spectrum rows computed once, written framenum times) against
    mfcc_forward + mfcc_stack_context bit for bit (both copy the same float32 spectrum values), on a
    ragged batch with one-frame, shorter-than-window, tile-sized and multi-tile utterances, and
    against the oracle's stacking of its own spectrum."""
    rng = np.random.default_rng(33)
    frames = [1, 2, 3, 8, 9, 16, 17, 40, 299, 304, 305, 700]
    lengths = [400 + 160 * (f - 1) - int(rng.integers(0, 160)) if f > 1 else 123 for f in frames]
    audios = [np.clip(rng.standard_normal(n) * 3000, -32768, 32767).astype(np.int16) for n in lengths]
    fe = M.MFCC(variant)
    lay = fe.layout(lengths)
    packed = torch.from_numpy(np.concatenate(audios)).cuda()
    spec = fe.forward(packed, lay, kind="logspec" if log else "spec")
    counts = np.diff(lay.frame_offsets)
    assert list(counts) == frames
    # every odd window up to 21 frames in one launch: rows of N * 257 floats start on every 16-byte phase when
    # N % 4 == 1 and N % 4 == 3 alike (the second kind reads the ring on 8-byte boundaries); 19 and 21 use the longer ring
    for framenum in (17, 5, 1, 7, 11, 15, 19, 21, 3, 9, 13):
        two = M.stack_context(spec, framenum, frame_counts=counts)
        one = M.spectrum_stack(fe, packed, lay, framenum, log=log)
        assert one.shape == (sum(frames), framenum * 257)
        assert torch.equal(one, two), framenum
    # beyond 21 frames the same call takes the two-kernel route
    assert torch.equal(M.spectrum_stack(fe, packed, lay, 23, log=log), M.stack_context(spec, 23, frame_counts=counts))
    # and the values are the reference's: spectrum of the oracle, stacked by the oracle
    u = 7                                                              # the 40-frame utterance
    ref = O.stack_context(O.spectrum257(audios[u], variant, log=log), 17)
    got = M.spectrum_stack(fe, packed, lay, 17, log=log)[lay.frame_offsets[u]:lay.frame_offsets[u + 1]].cpu().numpy()
    if log:
        assert np.abs(got - ref).max() < 2e-3        # ln of bins far below the row maximum, see test_log_spectrum_against_oracle
    else:
        assert (np.abs(got - ref) <= 2e-6 * np.abs(ref).max()).all()
    with pytest.raises(ValueError):
        M.spectrum_stack(fe, packed, lay, 17, out=torch.empty(3, 3, device="cuda"))


@pytest.mark.parametrize("variant", ["self5", "self15"])
def test_mfcc39_stack_one_launch_equals_two_kernels(M, variant):
    """mfcc_forward_stacked(MFCC_OUT_MFCC39): the finished MFCC-39 rows stacked over N frames in one launch
    (coeffs = 39, framenum = 11 of the reference's best model, MLP_5hidlayer_40.py:70-105,244-248) against
    mfcc_forward + mfcc_stack_context bit for bit, on a ragged batch with one-frame, shorter-than-window,
    tile-sized and multi-tile utterances, for every odd window 3 .. 21 (mfcc_stack_context itself is held to the
    reference's 39-wide, 11-frame golden in test_stacking_bit_exact), and against the oracle's stacking of its own rows."""
    rng = np.random.default_rng(34)
    frames = [1, 2, 3, 8, 9, 16, 17, 40, 299, 304, 305, 700, 5, 296, 297, 24]
    lengths = [400 + 160 * (f - 1) - int(rng.integers(0, 160)) if f > 1 else 123 for f in frames]
    audios = [np.clip(rng.standard_normal(n) * 3000, -32768, 32767).astype(np.int16) for n in lengths]
    fe = M.MFCC(variant)
    lay = fe.layout(lengths)
    packed = torch.from_numpy(np.concatenate(audios)).cuda()
    rows = fe.forward(packed, lay)
    counts = np.diff(lay.frame_offsets)
    assert list(counts) == frames
    for framenum in (11, 3, 5, 7, 9, 13, 15, 17, 19, 21):
        two = M.stack_context(rows, framenum, frame_counts=counts)
        one = M.mfcc39_stack(fe, packed, lay, framenum)
        assert one.shape == (sum(frames), framenum * 39)
        assert torch.equal(one, two), framenum
    assert torch.equal(M.mfcc39_stack(fe, packed, lay, 23), M.stack_context(rows, 23, frame_counts=counts))   # two-kernel route
    u = 7                                                              # the 40-frame utterance
    ref = O.stack_context(O.mfcc39(audios[u], variant, loop=False), 11)
    got = M.mfcc39_stack(fe, packed, lay, 11)[lay.frame_offsets[u]:lay.frame_offsets[u + 1]].cpu().numpy()
    assert np.abs(got - ref).max() < 1e-3
    # float32 samples, another numcep (run-time row width 3 * 12 = 36), the 8 kHz framing
    f32 = torch.from_numpy(np.concatenate(audios).astype(np.float32) / 32768).cuda()
    assert torch.equal(M.mfcc39_stack(fe, f32, lay, 11), M.stack_context(fe.forward(f32, lay), 11, frame_counts=counts))
    fe12 = M.MFCC(variant, numcep=12)
    assert torch.equal(M.mfcc39_stack(fe12, packed, lay, 7), M.stack_context(fe12.forward(packed, lay), 7, frame_counts=counts))
    fe8 = M.MFCC(variant, fs=8000)
    assert not fe8.plan.general
    l8 = [200 + 80 * (f - 1) - int(rng.integers(0, 80)) if f > 1 else 50 for f in frames]
    a8 = torch.from_numpy(np.concatenate([np.clip(rng.standard_normal(n) * 3000, -32768, 32767).astype(np.int16) for n in l8])).cuda()
    lay8 = fe8.layout(l8)
    assert torch.equal(M.mfcc39_stack(fe8, a8, lay8, 11), M.stack_context(fe8.forward(a8, lay8), 11, frame_counts=frames))
    with pytest.raises(ValueError):
        M.mfcc39_stack(fe, packed, lay, 11, out=torch.empty(3, 3, device="cuda"))


def test_mfcc39_stack_full_batch(M):
    """The bench batch (10 000 x 3 s) stacked over 11 frames in one launch: sampled utterances equal the
    two-kernel route bit for bit (tiles cut short at the end of the grid included)."""
    n_utt, n_len = 10000, 48000
    audio = (torch.randn(n_utt * n_len // 8, device="cuda").repeat(8) * 3000).clamp(-32768, 32767).to(torch.int16)
    fe = M.MFCC("self5")
    lay = fe.layout([n_len] * n_utt)
    got = M.mfcc39_stack(fe, audio, lay, 11)
    rows = fe.forward(audio, lay)
    for u in (0, 1, 4999, 9984, 9985, 9999):
        want = M.stack_context(rows[u * 299:(u + 1) * 299].contiguous(), 11)
        assert torch.equal(got[u * 299:(u + 1) * 299], want), u
    del got, rows


def test_spectrum_stack_other_instantiations(M):
    """The row writer of mfcc_forward_stacked for the other window lengths, float32 samples, the 200/80-sample
    framing with a 512-point FFT, a Hamming window, and a batch large enough that the layout cuts its last
    tiles short: always bit-equal to mfcc_forward + mfcc_stack_context."""
    rng = np.random.default_rng(5)
    frames = [33, 1, 7, 16, 299, 100, 17, 2, 640]
    lengths = [400 + 160 * (f - 1) - int(rng.integers(0, 160)) if f > 1 else 77 for f in frames]
    audio = np.concatenate([np.clip(rng.standard_normal(n) * 3000, -32768, 32767).astype(np.int16) for n in lengths])
    for kw, samples in ((dict(), torch.from_numpy(audio).cuda()),
                        (dict(), torch.from_numpy(audio.astype(np.float32) / 32768).cuda()),
                        (dict(window="hamming"), torch.from_numpy(audio).cuda())):
        fe = M.MFCC("self5", **kw)
        lay = fe.layout(lengths)
        spec = fe.forward(samples, lay, kind="logspec")
        counts = np.diff(lay.frame_offsets)
        for framenum in (9, 11, 17, 21):
            assert torch.equal(M.spectrum_stack(fe, samples, lay, framenum), M.stack_context(spec, framenum, frame_counts=counts))
    # 25 ms / 10 ms at 8 kHz with the 512-point FFT of the CTC scripts: 200/80-sample frames on the specialised kernel
    fe8 = M.MFCC("self5", fs=8000, num_fft=512)
    assert not fe8.plan.general
    l8 = [200 + 80 * (f - 1) - int(rng.integers(0, 80)) if f > 1 else 50 for f in frames]
    a8 = torch.from_numpy(np.concatenate([np.clip(rng.standard_normal(n) * 3000, -32768, 32767).astype(np.int16) for n in l8])).cuda()
    lay8 = fe8.layout(l8)
    assert list(np.diff(lay8.frame_offsets)) == frames
    spec8 = fe8.forward(a8, lay8, kind="spec")
    for framenum in (17, 7):
        assert torch.equal(M.spectrum_stack(fe8, a8, lay8, framenum, log=False), M.stack_context(spec8, framenum, frame_counts=frames))
    # 800 equal utterances: 32 tiles beyond one round of 3 x 256 workgroups are cut into short ones
    fe = M.MFCC("self15")
    n_utt, n_len = 800, 16000
    big = (torch.randn(n_utt * n_len, device="cuda") * 3000).clamp(-32768, 32767).to(torch.int16)
    layb = fe.layout([n_len] * n_utt)
    specb = fe.forward(big, layb, kind="spec")
    assert torch.equal(M.spectrum_stack(fe, big, layb, 17, log=False), M.stack_context(specb, 17, frame_counts=np.diff(layb.frame_offsets)))


def test_corpus_pipeline_statistics_per_chunk_or_at_the_end(M):
    """corpus_mfcc_cmvn: the statistics pass after every chunk or once over the resident rows -- same
    count, sums equal to float64 rounding, normalised rows equal to float32 rounding of the constants."""
    rng = np.random.default_rng(77)
    lengths = [int(v) for v in rng.integers(1500, 9000, size=50)]
    audio = torch.from_numpy(np.concatenate([np.clip(rng.standard_normal(n) * 3000, -32768, 32767).astype(np.int16) for n in lengths])).cuda()
    fe = M.MFCC("self5")
    a = M.corpus_mfcc_cmvn(fe, audio, lengths, chunk_utts=7)
    b = M.corpus_mfcc_cmvn(fe, audio, lengths, chunk_utts=7, per_chunk_sums=True)
    c = M.corpus_mfcc_cmvn(fe, audio, lengths, chunk_utts=1000, static_scratch=True)
    for other in (b, c):
        assert other[1][0] == a[1][0]
        assert np.allclose(other[1].cpu().numpy(), a[1].cpu().numpy(), rtol=1e-12, atol=1e-7)
        assert (other[0] - a[0]).abs().max() < 1e-5
    assert torch.equal(a[0], c[0])    # one pass over the same rows in the same order: same bits, whatever the chunking
    # the per-chunk passes on a second stream, beside the next chunk's extraction: the bits of the per-chunk passes in line
    for _ in range(3):
        o = M.corpus_mfcc_cmvn(fe, audio, lengths, chunk_utts=7, overlap=True)
        assert torch.equal(o[1], b[1]) and torch.equal(o[0], b[0])
    o = M.corpus_mfcc_cmvn(fe, audio, lengths, chunk_utts=7, overlap=True, static_scratch=True)
    assert torch.equal(o[1], b[1]) and torch.equal(o[0], b[0])


@pytest.mark.parametrize("variant,numcep,dtype", [("self5", 13, np.int16), ("self15", 13, np.float32), ("self5", 12, np.int16), ("self5", 16, np.int16)])
def test_forward_sums_rows_and_statistics(M, variant, numcep, dtype):
    """mfcc_forward_sums: rows bit-identical to mfcc_forward, the sums equal to a float64 reduction of those rows to
    float32 rounding of a tile's partial sums, the same bits on every run, accumulation over calls; ragged batch with
    one-frame utterances and tiles that end inside an utterance."""
    rng = np.random.default_rng(123)
    lengths = [400, 401, 1, 5000, 48000, 160 * 300 + 400, 7777, 560, 16000 * 7 + 13, 3000] + [int(v) for v in rng.integers(500, 60000, size=30)]
    audios = [np.clip(rng.standard_normal(n) * 3000, -32768, 32767).astype(np.int16) for n in lengths]
    packed = np.concatenate(audios)
    audio = torch.from_numpy(packed if dtype == np.int16 else packed.astype(np.float32)).cuda()
    fe = M.MFCC(variant, numcep=numcep)
    lay = fe.layout(lengths)
    d = 3 * numcep
    ref_rows = fe.forward(audio, lay)
    sums = torch.zeros(1 + 2 * d, dtype=torch.float64, device="cuda")
    rows = fe.forward_sums(audio, lay, sums)
    assert torch.equal(rows, ref_rows)
    want = M.cmvn_accumulate(ref_rows).cpu().numpy()
    got = sums.cpu().numpy()
    assert got[0] == want[0] == lay.total_frames
    # float32 partial sums of up to 296 values per tile: a few 1e-7 of the sum of the magnitudes
    x = ref_rows.cpu().numpy().astype(np.float64)
    mag = np.concatenate([np.abs(x).sum(0), (x * x).sum(0)])
    assert np.all(np.abs(got[1:] - want[1:]) <= 2e-6 * mag + 1e-9), np.max(np.abs(got[1:] - want[1:]) / (mag + 1e-30))
    again = torch.zeros_like(sums)
    ws = torch.empty(fe.forward_sums_workspace(lay) + 5, dtype=torch.float32, device="cuda")
    fe.forward_sums(audio, lay, again, out=torch.empty_like(rows), workspace=ws)
    assert torch.equal(again, sums)
    fe.forward_sums(audio, lay, again, workspace=ws)   # accumulates into what is there
    assert np.allclose(again.cpu().numpy(), 2 * got, rtol=1e-15)
    # normalisation constants from either set of sums: the same to float32 rounding
    mm_a, vv_a = M.cmvn_constants(sums)
    mm_b, vv_b = M.cmvn_constants(torch.from_numpy(want).cuda())
    assert torch.allclose(mm_a, mm_b, rtol=1e-5, atol=1e-5) and torch.allclose(vv_a, vv_b, rtol=1e-5)


def test_forward_sums_refusals_and_corpus_pipeline(M):
    rng = np.random.default_rng(5)
    lengths = [int(v) for v in rng.integers(1500, 9000, size=50)]
    audio = torch.from_numpy(np.concatenate([np.clip(rng.standard_normal(n) * 3000, -32768, 32767).astype(np.int16) for n in lengths])).cuda()
    fe = M.MFCC("self5")
    a = M.corpus_mfcc_cmvn(fe, audio, lengths, chunk_utts=7)
    f = M.corpus_mfcc_cmvn(fe, audio, lengths, chunk_utts=7, fused_sums=True)
    g = M.corpus_mfcc_cmvn(fe, audio, lengths, chunk_utts=1000, fused_sums=True)
    assert f[1][0] == a[1][0] == g[1][0]
    assert np.allclose(f[1].cpu().numpy(), a[1].cpu().numpy(), rtol=3e-6, atol=1e-4)
    assert (f[0] - a[0]).abs().max() < 1e-4 and (g[0] - a[0]).abs().max() < 1e-4
    gen = M.MFCC("self5", kernel="general")
    lay = gen.layout(lengths[:3])
    with pytest.raises(RuntimeError):
        gen.forward_sums(audio[:sum(lengths[:3])], lay, torch.zeros(79, dtype=torch.float64, device="cuda"))
    with pytest.raises(ValueError):
        M.corpus_mfcc_cmvn(fe, audio, lengths, fused_sums=True, overlap=True)
    # silence: ln 0 = -inf rows make the sums non-finite exactly as the two-pass route does
    z = torch.zeros(4000, dtype=torch.int16, device="cuda")
    lz = fe.layout([4000])
    s1 = torch.zeros(79, dtype=torch.float64, device="cuda")
    rz = fe.forward_sums(z, lz, s1)
    s2 = M.cmvn_accumulate(rz)
    assert torch.equal(torch.isfinite(s1), torch.isfinite(s2))
