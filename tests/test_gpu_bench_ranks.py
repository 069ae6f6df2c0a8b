"""bench.py's own N-rank path on one GPU: the hand-started launcher (torch.distributed.run relay, free
rendezvous port), process-group set-up, barriers, the MAX over ranks and rank 0's JSON line, with two
ranks sharing the GPU over gloo.  The headline config and the corpus + CMVN config both run as fresh
child processes of the fork server (never of this process, which holds a HIP context)."""
import json
import multiprocessing as mp
import os
import sys

import pytest

from rank_worker import run_command

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*flags):
    ctx = mp.get_context("forkserver")
    q = ctx.Queue()
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), *flags]
    p = ctx.Process(target=run_command, args=(cmd, ROOT, q))
    p.start()
    rc, out, err = q.get(timeout=1000)
    p.join(60)
    assert rc == 0, f"rc {rc}\n{out[-2000:]}\n{err[-4000:]}"
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out          # rank 0 alone prints, one line
    return json.loads(lines[0])


@pytest.mark.timeout(1200)
def test_bench_headline_two_ranks_over_gloo():
    rec = _bench("--gpus", "2", "--backend", "gloo", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-extra")
    assert rec["n_gpus"] == 2 and rec["steps"] == 2 and rec["warmup"] == 1 and rec["scaling"] == "weak"
    assert rec["config"]["frames_per_gpu"] == 10000 * 299
    # whole-job value: both ranks' frames over the slowest rank's time
    assert abs(rec["value"] - 2 * 10000 * 299 / (rec["ms_per_step"] * 1e-3)) <= 1e-3 * rec["value"]
    assert rec["roofline"]["kernel_ms"] <= rec["ms_per_step"] * 1.001
    assert rec["preconditioning"]["preconditioning_launches"] >= 20


@pytest.mark.timeout(1200)
def test_bench_corpus_cmvn_two_ranks_over_gloo():
    rec = _bench("--config", "D", "--gpus", "2", "--hours", "0.5", "--backend", "gloo", "--steps", "2", "--warmup", "1")
    n_total = 600                        # 0.5 h of three-second utterances
    assert rec["n_gpus"] == 2 and rec["scaling"] == "strong"
    assert rec["config"]["frames_total"] == n_total * 299
    assert rec["config"]["utterances_rank0"] == n_total // 2
    assert rec["checks"]["frames_in_sums"] == n_total * 299      # the sums cover the whole corpus, not rank 0's shard
    assert rec["checks"]["oracle_checked"] is True
