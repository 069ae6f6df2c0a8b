#!/usr/bin/env python3
"""Headline benchmark: MFCC-39 frames/s on synthetic 16 kHz audio (BASELINE.json configs[1]).

    python bench.py [--gpus N] [--steps K] [--warmup W]

One "step" is one pass of the fused hot path (mfcc_forward, MFCC-39 output) over a batch of
10 000 three-second int16 utterances that is already resident in HBM; the [2 990 000, 39]
float32 result is written to HBM.  With N > 1 (one process per GPU, launched by
torch.distributed.run) every rank processes its own 10 000-utterance shard -- utterances are
independent, so there is no collective on the data path (weak scaling); ranks meet only at
the barriers around the timed region.

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  roofline      the fused kernel against the HBM roofline: algorithmic bytes
                (2 B/sample read + 156 B/frame written) / mean launch duration (HIP events on
                the launch stream) / 8 TB/s;
  cpu_baseline  the loop-faithful numpy port of the reference (oracle/mfcc_oracle.py, the
                checker -- never the product) timed on this host's cores on a bounded sample.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

N_UTT = 10000
UTT_LEN = 48000            # 3 s at 16 kHz
FRAMES_PER_UTT = 299       # 1 + ceil((48000 - 400) / 160)
HBM_PEAK_GBS = 8000.0      # MI355X HBM3E peak


def _oracle_job(seed_and_count):
    """Worker of the CPU baseline: loop-faithful numpy MFCC-39 on `count` synthetic utterances."""
    import numpy as np
    from oracle import mfcc_oracle as O
    seed, count = seed_and_count
    rng = np.random.default_rng(seed)
    frames = 0
    t0 = time.perf_counter()
    for _ in range(count):
        audio = np.clip(rng.standard_normal(UTT_LEN) * 3000, -32768, 32767).astype(np.int16)
        frames += O.mfcc39(audio, "self5", loop=True).shape[0]
    return frames, time.perf_counter() - t0


def cpu_baseline():
    """~20 s of CPU work: ~6 s on 1 core first, then ~10 s per worker of a process pool on the box's CPU share."""
    import multiprocessing as mp
    f1, t1 = _oracle_job((1, 8))                      # calibrate (also warms caches)
    per_utt = t1 / 8
    n1 = max(8, min(4000, int(6.0 / per_utt)))
    f1, t1 = _oracle_job((2, n1))
    cores = max(1, min(16, os.cpu_count() or 1))
    per_worker = max(4, min(4000, int(10.0 / per_utt)))
    ctx = mp.get_context("spawn")
    with ctx.Pool(cores) as pool:
        pool.map(_oracle_job, [(100 + i, 1) for i in range(cores)])   # start-up outside the clock
        t0 = time.perf_counter()
        res = pool.map(_oracle_job, [(200 + i, per_worker) for i in range(cores)])
        wall = time.perf_counter() - t0
    frames = sum(r[0] for r in res)
    return {
        "value": round(frames / wall, 1), "unit": "frames/s", "cores": cores, "kind": "port",
        "sample": f"{cores * per_worker} synthetic 3 s utterances ({frames} frames) over {cores} processes, "
                  f"loop-faithful float64 numpy restatement of mfcc-self5.py",
        "value_1core": round(f1 / t1, 1), "sample_1core": f"{n1} utterances ({f1} frames), 1 process",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--variant", default="self5", choices=("self5", "self15"))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        # started by hand without a launcher: start one process per GPU and relay the result
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", "29531", os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))

    cpu = None
    if world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline()          # before the GPU is touched: the pool must not inherit a HIP context

    import torch
    import mfcc_amd

    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)

    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    audio = torch.empty((N_UTT, UTT_LEN), dtype=torch.int16, device=dev)
    for i in range(0, N_UTT, 1000):   # generate in chunks: the float32 noise is 4x the int16 batch
        audio[i:i + 1000] = (torch.randn(1000, UTT_LEN, generator=gen, device=dev) * 3000).clamp(-32768, 32767).to(torch.int16)
    fe = mfcc_amd.MFCC(args.variant, device=dev)
    layout = fe.layout([UTT_LEN] * N_UTT)
    frames = layout.total_frames
    assert frames == N_UTT * FRAMES_PER_UTT
    out = torch.empty((frames, 39), dtype=torch.float32, device=dev)
    flat = audio.view(-1)

    def step():
        fe.forward(flat, layout, out=out, kind="mfcc39")

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    starts = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    ends = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for i in range(args.steps):
        starts[i].record()            # same stream as the kernel launch (torch's current stream)
        step()
        ends[i].record()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    kernel_ms = sum(s.elapsed_time(e) for s, e in zip(starts, ends)) / args.steps

    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # the one collective of the front-end, outside the timed region: corpus CMVN sums
        sums = mfcc_amd.cmvn_allreduce(mfcc_amd.cmvn_accumulate(out))
        assert int(sums[0].item()) == frames * world

    assert torch.isfinite(out[:: 9973]).all()
    if rank == 0:
        total_frames = frames * world
        value = total_frames * args.steps / elapsed
        bytes_per_launch = 2 * N_UTT * UTT_LEN + 4 * 39 * frames          # int16 in + fp32 out
        achieved = bytes_per_launch / (kernel_ms * 1e-3) / 1e9
        traffic = traffic_bytes = None
        tpath = os.path.join(ROOT, "profiles", "r1_hbm_traffic.json")
        if os.path.exists(tpath) and args.variant == "self5":
            # HBM bytes per launch of this very workload, from separate rocprofv3 --pmc passes
            # (FETCH_SIZE doubled as gfx950 requires, + WRITE_SIZE); counters cannot run inside this process
            with open(tpath) as fh:
                traffic_bytes = json.load(fh)["hbm_bytes_per_launch"]
            traffic = round(traffic_bytes / (kernel_ms * 1e-3) / 1e9, 1)
        rec = {
            "metric": "MFCC-39 frames/sec (whole node)",
            "value": round(value, 1),
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"{N_UTT} x 3 s synthetic 16 kHz int16 utterances per GPU -> MFCC-39 "
                            f"(BASELINE.json configs[1]), variant {args.variant}",
                "frames_per_gpu": frames, "frame": "400/160 samples, 512-pt FFT, 26 mel, 13 cep + d + dd",
                "input": "int16 resident in HBM", "output": "float32 [frames, 39] in HBM",
                "parallelism": f"utterance-sharded x{world}, no data-path collective",
            },
            "audio_hours_per_s": round(value / 360000.0, 2),
            "x_realtime": round(value / 100.0, 1),
            "roofline": {
                "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                "kernel": "mfcc::fused_kernel<0, 13, short, mfcc::FrameGeom<400, 160>>", "kernel_ms": round(kernel_ms, 4),
                "algorithmic_bytes_per_launch": bytes_per_launch,
                "traffic_bytes_per_launch": traffic_bytes,
                "traffic_source": "profiles/r1_hbm_traffic.json (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE)",
            },
        }
        if cpu is not None:
            rec["cpu_baseline"] = cpu
        print(json.dumps(rec), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
