"""N > 1 path on CPU: two gloo ranks shard a ragged corpus by utterance, accumulate CMVN sums on
their shards and all-reduce them with the product's cmvn_allreduce; the result must equal the
single-process sums.  (The per-shard sums come from the oracle here because there is no GPU;
what is under test is the sharding and the collective.)"""
import os
import socket

import numpy as np
import pytest

torch = pytest.importorskip("torch")
import torch.distributed as dist  # noqa: E402
import torch.multiprocessing as mp  # noqa: E402


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, lengths, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mfcc_amd.frontend import cmvn_allreduce, cmvn_constants, shard_utterances
    from oracle import mfcc_oracle as O
    begin, end = shard_utterances(lengths, world)[rank]
    rng_all = [np.random.default_rng(1000 + u) for u in range(len(lengths))]
    feats = [O.mfcc39(np.clip(rng_all[u].standard_normal(lengths[u]) * 3000, -32768, 32767).astype(np.int16),
                      "self5", loop=False) for u in range(begin, end)]
    x = np.concatenate(feats) if feats else np.zeros((0, 39))
    n, s, ss = O.cmvn_sums(x)
    sums = torch.from_numpy(np.concatenate([[float(n)], s, ss]))
    sums = cmvn_allreduce(sums)
    mm, vv = cmvn_constants(sums)
    local = torch.from_numpy(np.concatenate([[float(n)], s, ss]))
    q.put((rank, begin, end, sums.numpy().copy(), mm.numpy().copy(), vv.numpy().copy(), local.numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_cmvn_allreduce_matches_single_process():
    from oracle import mfcc_oracle as O
    lengths = [4000, 16000, 800, 48000, 1200, 9000, 24000, 400]
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, lengths, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res[0][1] == 0 and res[0][2] == res[1][1] and res[1][2] == len(lengths)
    feats = [O.mfcc39(np.clip(np.random.default_rng(1000 + u).standard_normal(n) * 3000, -32768, 32767).astype(np.int16),
                      "self5", loop=False) for u, n in enumerate(lengths)]
    n, s, ss = O.cmvn_sums(np.concatenate(feats))
    for r in res:
        sums = r[3]
        assert sums[0] == n
        assert np.allclose(sums[1:40], s, rtol=1e-12, atol=1e-9) and np.allclose(sums[40:], ss, rtol=1e-12)
    assert np.array_equal(res[0][3], res[1][3])          # every rank holds identical constants
    # the combination is an all-gather plus a sum in rank order: bit-equal to adding the per-rank
    # vectors in that order in one process
    ordered = res[0][6].copy()
    ordered += res[1][6]
    assert np.array_equal(res[0][3], ordered)
    rmm, rvv = O.cmvn_constants(n, s, ss)
    assert np.allclose(res[0][4], rmm, rtol=1e-6) and np.allclose(res[0][5], rvv, rtol=1e-6)
