"""The corpus pipeline of BASELINE configs[3] at reduced size on N ranks, N = 1 and 2 (more where the
box has more GPUs): every rank extracts MFCC-39 for its utterance shard with the HIP kernels,
accumulates CMVN sums, exchanges them (RCCL where every rank has its own GPU, gloo where ranks have
to share one) and normalises its rows in place.  Checked: all ranks hold bit-identical sums, the sums
do not depend on the number of ranks beyond float64 rounding, and the normalised rows of a strided
sample of utterances equal a float64 numpy CMVN of the oracle's features."""
import multiprocessing as mp
import socket

import numpy as np
import pytest
import torch

from oracle import mfcc_oracle as O
from rank_worker import corpus_audio, run_rank

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worlds():
    n = torch.cuda.device_count() if torch.cuda.is_available() else 1
    return sorted({1, 2, min(max(n, 1), 8)})


@pytest.fixture(scope="module")
def corpus():
    rng = np.random.default_rng(5)
    lengths = rng.integers(1500, 30000, size=400).astype(np.int64)
    audios = corpus_audio(lengths)
    feats = [O.mfcc39(a, "self5", loop=False) for a in audios]
    n, s, ss = O.cmvn_sums(np.concatenate(feats))
    mm, vv = O.cmvn_constants(n, s, ss)
    return lengths, feats, (n, s, ss), (mm, vv)


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world", _worlds())
def test_corpus_cmvn_on_n_ranks(corpus, world):
    lengths, feats, (n, s, ss), (mm, vv) = corpus
    sample = list(range(0, len(lengths), 37))
    # children come from a fork server that was started before this process touched the GPU
    ctx = mp.get_context("forkserver")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=run_rank, args=(r, world, port, lengths, sample, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=500) for _ in range(world)), key=lambda r: r[0])
    for p in procs:
        p.join(120)
    for r in res:
        assert r[8] is None, r[8]
    assert res[0][1] == 0 and res[-1][2] == len(lengths) and all(res[i][2] == res[i + 1][1] for i in range(world - 1))
    for r in res[1:]:
        assert np.array_equal(r[4], res[0][4])            # identical bits on every rank
    sums = res[0][4]
    assert sums[0] == n
    # float32 features against the float64 oracle: sums agree to the features' own tolerance
    assert np.allclose(sums[1:40] / n, s / n, rtol=0, atol=1e-3)
    assert np.allclose(sums[40:] / n, ss / n, rtol=1e-4)
    seen = 0
    for r in res:
        for u, rows in r[7].items():
            ref = O.cmvn_apply(feats[u], mm, vv)
            assert rows.shape == ref.shape
            np.testing.assert_allclose(rows, ref, rtol=0, atol=1e-3)
            seen += 1
    assert seen == len(sample)
    print(f"[corpus] world {world}, exchange over {res[0][3]}")
