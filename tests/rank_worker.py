"""Child process of tests/test_gpu_corpus_ranks.py: one rank of the corpus pipeline (MFCC-39 of its
utterance shard -> CMVN sums -> exchange -> in-place normalisation) on a GPU."""
import os

import numpy as np


def corpus_audio(lengths, seed=77):
    rng = np.random.default_rng(seed)
    return [np.clip(rng.standard_normal(int(n)) * (500 + 40 * (i % 50)), -32768, 32767).astype(np.int16) for i, n in enumerate(lengths)]


def run_rank(rank, world, port, lengths, sample_utts, q):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        import torch
        import torch.distributed as dist
        import mfcc_amd
        ndev = torch.cuda.device_count()
        dev = torch.device("cuda", rank % ndev)
        torch.cuda.set_device(dev)
        backend = "nccl" if ndev >= world else "gloo"       # fewer GPUs than ranks: ranks share a GPU, sums travel over gloo
        if world > 1:
            dist.init_process_group(backend, rank=rank, world_size=world, **({"device_id": dev} if backend == "nccl" else {}))
        fe = mfcc_amd.MFCC("self5", device=dev)
        begin, end = mfcc_amd.shard_utterances(lengths, world, fe.plan)[rank]
        audios = corpus_audio(lengths)[begin:end]
        packed = torch.from_numpy(np.concatenate(audios)).to(dev) if audios else torch.zeros(0, dtype=torch.int16, device=dev)
        feats, sums, mm, vv = mfcc_amd.corpus_mfcc_cmvn(fe, packed, lengths[begin:end], chunk_utts=97)
        fo = np.concatenate([[0], np.cumsum([fe.plan.frames_for_length(int(n)) for n in lengths[begin:end]])])
        rows = {int(u): feats[fo[u - begin]:fo[u - begin + 1]].cpu().numpy() for u in sample_utts if begin <= u < end}
        q.put((rank, begin, end, backend, sums.cpu().numpy(), mm.cpu().numpy(), vv.cpu().numpy(), rows, None))
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
    except Exception as exc:  # report instead of hanging the parent
        import traceback
        q.put((rank, 0, 0, "", None, None, None, None, traceback.format_exc() + repr(exc)))


def run_command(cmd, cwd, q):
    """Runs a command as a child of the fork server (a process that never held a HIP context, so starting
    a program from it is allowed) and reports (exit code, stdout, stderr)."""
    import subprocess
    try:
        r = subprocess.run(cmd, cwd=cwd, capture_output=True, text=True, timeout=900)
        q.put((r.returncode, r.stdout, r.stderr))
    except Exception as exc:
        q.put((-1, "", repr(exc)))
