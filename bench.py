#!/usr/bin/env python3
"""Headline benchmark: MFCC-39 frames/s on synthetic 16 kHz audio (BASELINE.json configs[1]).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python bench.py --config D [--gpus N] [--hours H]      corpus + CMVN pipeline (BASELINE.json configs[3])

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
        "value": round(frames / wall, 1), "unit": "frames/s", "cores": cores, "kind": "port", "cpu_model": cpu_model(),
        "sample": f"{cores * per_worker} synthetic 3 s utterances ({frames} frames) over {cores} processes, "
                  f"loop-faithful float64 numpy restatement of mfcc-self5.py",
        "value_1core": round(f1 / t1, 1), "sample_1core": f"{n1} utterances ({f1} frames), 1 process",
    }


def cpu_model():
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def free_port():
    """A TCP port nobody listens on right now, for the rendezvous of a hand-started N-rank run."""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sock:
        sock.bind(("127.0.0.1", 0))
        return sock.getsockname()[1]


def max_over_ranks(value, dist, torch, dev):
    """MAX of a host float over the ranks (device tensor for RCCL, host tensor for gloo)."""
    t = torch.tensor([value], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def kernel_revision():
    """Identifies the kernel sources a measurement belongs to: sha1 over mfcc_amd/csrc/*."""
    import glob
    import hashlib
    h = hashlib.sha1()
    for path in sorted(glob.glob(os.path.join(ROOT, "mfcc_amd", "csrc", "*"))):
        with open(path, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:12]


def synth_audio(n_utt, dev, gen, torch):
    audio = torch.empty((n_utt, UTT_LEN), dtype=torch.int16, device=dev)
    for i in range(0, n_utt, 1000):   # generate in chunks: the float32 noise is 4x the int16 batch
        m = min(1000, n_utt - i)
        audio[i:i + m] = (torch.randn(m, UTT_LEN, generator=gen, device=dev) * 3000).clamp(-32768, 32767).to(torch.int16)
    return audio


def precondition(step, torch, cap=100, window=10):
    """Untimed clock-ramp phase, separate from (and in front of) the --warmup launches: a cold part
    raises its clocks over the first few dozen launches of this kernel, and a short run would time the
    ramp instead of the kernel.  Launches run in windows of `window`; the phase ends when the median
    launch times of three windows in a row agree within 1 %, or after `cap` launches."""
    medians, first, last, n = [], None, None, 0
    while n < cap:
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(window)]
        for a, b in evs:
            a.record()
            step()
            b.record()
        torch.cuda.synchronize()
        w = [a.elapsed_time(b) for a, b in evs]
        n += window
        first = w[0] if first is None else first
        last = w[-1]
        medians.append(sorted(w)[len(w) // 2])
        if len(medians) >= 3 and max(medians[-3:]) - min(medians[-3:]) <= 0.01 * min(medians[-3:]):
            break
    return {"preconditioning_launches": n, "first_launch_ms": round(first, 4), "last_launch_ms": round(last, 4),
            "window_median_ms": [round(m, 4) for m in medians],
            "note": "untimed clock-ramp launches in front of the --warmup launches; not part of steps, warmup or value"}


def check_config_c(torch, mfcc_amd, fe, audio, ctx, n_check=16):
    """Rows of the one-launch stacked output against the two-kernel route (spectrum rows of the same
    utterances, then stack_context) on n_check utterances spread over the batch, first and last included."""
    import numpy as np
    picks = sorted(set(np.linspace(0, N_UTT - 1, n_check).astype(int).tolist()))
    sub = audio[picks].contiguous().view(-1)
    lay = fe.layout([UTT_LEN] * len(picks))
    spec = fe.forward(sub, lay, kind="logspec")
    want = mfcc_amd.stack_context(spec, 17, frame_counts=[FRAMES_PER_UTT] * len(picks))
    got = torch.cat([ctx[u * FRAMES_PER_UTT:(u + 1) * FRAMES_PER_UTT] for u in picks])
    same = bool(torch.equal(got, want)) and bool(torch.isfinite(got).all())
    return same, f"{len(picks)} utterances x {FRAMES_PER_UTT} rows x 4369 floats bit-equal to forward(logspec) + stack_context"


def extra_measurements(torch, mfcc_amd, dev, audio, fe, layout):
    """Compact figures for the other BASELINE configurations, carried in the same JSON line:
    C = 257-bin ln power spectrum with 17-frame context stacking of the same 10 000 utterances,
    E = host-observed latency of one 10 ms hop of B lock-step live streams."""
    import numpy as np
    extra = {}
    frames = layout.total_frames
    try:
        ctx = torch.empty((frames, 17 * 257), dtype=torch.float32, device=dev)
        flat = audio.view(-1)

        def run_c():
            mfcc_amd.spectrum_stack(fe, flat, layout, 17, out=ctx)
        for _ in range(2):
            run_c()
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(5)]
        for a, b in ev:
            a.record()
            run_c()
            b.record()
        torch.cuda.synchronize()
        ms = sorted(a.elapsed_time(b) for a, b in ev)[2]
        alg = 2 * N_UTT * UTT_LEN + 4 * 257 * 17 * frames
        checked, how = check_config_c(torch, mfcc_amd, fe, audio, ctx)
        extra["C"] = {"workload": "10000 x 3 s -> ln |X|^2/512 [frames, 257] -> 17-frame stack [frames, 4369] (BASELINE.json configs[2])",
                      "ms": round(ms, 3), "frames_per_s": round(frames / (ms * 1e-3), 1), "algorithmic_bytes": alg,
                      "hbm_frac": round(alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "checked": checked, "check": how}
        del ctx
        torch.cuda.empty_cache()
    except Exception as exc:   # a secondary figure must not take the headline down
        extra["C"] = {"error": repr(exc)[:200]}
    try:
        e = {}
        for B in (1, 64):
            s = mfcc_amd.StreamingMFCC("self5", n_streams=B, max_chunk=160, device=dev)
            hops = 520
            chunk = (torch.randn(B, 160 * hops, device=dev) * 3000).clamp(-32768, 32767).to(torch.int16)
            out = torch.empty((B, s.max_rows, 39), dtype=torch.float32, device=dev)
            lat = []
            for h in range(hops):
                t0 = time.perf_counter()
                s.push(chunk[:, h * 160:(h + 1) * 160], out=out)
                torch.cuda.synchronize()
                if h >= 20:
                    lat.append((time.perf_counter() - t0) * 1e6)
            lat.sort()
            e[f"B{B}"] = {"us_per_hop_p50": round(lat[len(lat) // 2], 1), "us_per_hop_p99": round(lat[int(len(lat) * 0.99)], 1)}
        e["workload"] = "10 ms hops of B lock-step int16 streams, push + host wait per hop (BASELINE.json configs[4])"
        extra["E"] = e
    except Exception as exc:
        extra["E"] = {"error": repr(exc)[:200]}
    return extra


def check_config_d(torch, feats, audio, n_local, mm, vv, variant, world, n_check=8):
    """Checks of the corpus pipeline's result, outside the clock.  (1) Sampled utterances of this rank
    against the CPU oracle (the checker, never the product): normalised rows vs (oracle rows + mm) * vv,
    bound 1e-3 * vv per column as for the un-normalised rows.  (2) Every row of this rank's shard:
    float64 column moments of the normalised rows (exactly 0 and 1 on one rank; with more ranks the
    shard's own moments only approximate the corpus constants)."""
    import numpy as np
    from oracle import mfcc_oracle as O
    picks = sorted(set(np.linspace(0, n_local - 1, n_check).astype(int).tolist()))
    mm64, vv64 = mm.double().cpu().numpy(), vv.double().cpu().numpy()
    worst = 0.0
    ok = True
    for u in picks:
        a = audio[u * UTT_LEN:(u + 1) * UTT_LEN].cpu().numpy()
        want = (O.mfcc39(a, variant, loop=False) + mm64) * vv64
        got = feats[u * FRAMES_PER_UTT:(u + 1) * FRAMES_PER_UTT].double().cpu().numpy()
        err = np.abs(got - want) / vv64
        worst = max(worst, float(err.max()))
        ok = ok and bool((err <= 1e-3 + 1e-6 * np.abs(want / vv64)).all())
    s1 = torch.zeros(feats.shape[1], dtype=torch.float64, device=feats.device)
    s2 = torch.zeros_like(s1)
    for i in range(0, feats.shape[0], 1 << 22):
        x = feats[i:i + (1 << 22)].double()
        s1 += x.sum(0)
        s2 += (x * x).sum(0)
    n = feats.shape[0]
    mean = s1 / n
    var = s2 / n - mean * mean
    res = {"oracle_checked": ok, "oracle_sample": f"{len(picks)} utterances ({len(picks) * FRAMES_PER_UTT} rows) of rank 0",
           "oracle_max_abs_err_unnormalised_units": round(worst, 7),
           "normalised_mean_abs_max_all_rows_rank0": float(mean.abs().max().item()),
           "normalised_var_minus_1_abs_max_all_rows_rank0": float((var - 1).abs().max().item())}
    if world == 1:
        ok_m = res["normalised_mean_abs_max_all_rows_rank0"] < 1e-4 and res["normalised_var_minus_1_abs_max_all_rows_rank0"] < 1e-3
        res["moments_checked"] = bool(ok_m)
        assert ok_m, res
    assert ok, res
    return res


def run_config_d(args, rank, world, dev, torch, mfcc_amd, dist):
    """BASELINE.json configs[3]: hours of synthetic audio utterance-sharded over the GPUs, MFCC-39 kept in
    HBM, corpus-wide CMVN (float64 sums, one all-gather of 79 doubles over RCCL), rows normalised in place."""
    import numpy as np
    n_total = int(round(args.hours * 3600 / 3.0))                       # three-second utterances in the corpus
    lengths_all = np.full(n_total, UTT_LEN, dtype=np.int64)
    fe = mfcc_amd.MFCC(args.variant, device=dev)
    begin, end = mfcc_amd.shard_utterances(lengths_all, world, fe.plan)[rank]
    n_local = end - begin
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    audio = synth_audio(n_local, dev, gen, torch).view(-1)
    lengths = lengths_all[begin:end]
    frames_local = n_local * FRAMES_PER_UTT
    out = torch.empty((frames_local, 39), dtype=torch.float32, device=dev)
    # warm-up on the first chunk: loads every kernel and the collective
    for _ in range(max(1, args.warmup)):
        mfcc_amd.corpus_mfcc_cmvn(fe, audio[:min(n_local, 2000) * UTT_LEN], lengths[:2000], out=out[:min(n_local, 2000) * FRAMES_PER_UTT])
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    phase_ms = {}
    step_events = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        events = {}
        feats, sums, mm, vv = mfcc_amd.corpus_mfcc_cmvn(fe, audio, lengths, chunk_utts=10000, out=out, events=events,
                                                        overlap=args.overlap, fused_sums=args.fused_sums and not args.overlap)
        step_events.append(events)   # read after the clock stops: querying ~70 event pairs takes about 2 ms of host time
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    for events in step_events:
        for k, evs in events.items():
            phase_ms[k] = phase_ms.get(k, 0.0) + sum(a.elapsed_time(b) for a, b in evs)
    n_in_sums = int(sums[0].item())
    checks = check_config_d(torch, feats, audio, n_local, mm, vv, args.variant, world) if rank == 0 else None
    if world > 1:
        elapsed = max_over_ranks(elapsed, dist, torch, dev)
    assert n_in_sums == n_total * FRAMES_PER_UTT, (n_in_sums, n_total * FRAMES_PER_UTT)
    if rank == 0:
        total_frames = n_total * FRAMES_PER_UTT
        value = total_frames * args.steps / elapsed
        alg_rank = 2 * n_local * UTT_LEN + 468 * frames_local            # pass 1: 2 B/sample + 156 B/frame; pass 2: 2 x 156 B/frame
        step_s = elapsed / args.steps
        rec = {
            "metric": "MFCC-39 + corpus CMVN frames/sec (whole node)", "value": round(value, 1), "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(step_s * 1e3, 3),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.hours:g} h of synthetic 16 kHz int16 audio ({n_total} x 3 s) utterance-sharded over {world} GPU(s) "
                                   f"-> MFCC-39 resident in HBM -> corpus CMVN (BASELINE.json configs[3]), variant {args.variant}",
                       "utterances_rank0": n_local, "frames_total": total_frames,
                       "parallelism": f"utterance-sharded x{world}; one all-gather of 79 float64 per rank (RCCL), summed in rank order"},
            "audio_hours_per_s": round(value / 360000.0, 2), "x_realtime": round(value / 100.0, 1),
            "phases_ms_rank0_per_step": {k: round(v / args.steps, 3) for k, v in phase_ms.items()},
            "roofline": {"bound": "hbm", "achieved": round(alg_rank / step_s / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(alg_rank / step_s / 1e9 / HBM_PEAK_GBS, 4), "traffic": None,
                         "algorithmic_bytes_per_step_rank0": alg_rank,
                         "note": "two passes over the features: 2 B/sample + 156 B/frame written, then 156 B/frame read and written"},
            "statistics": "fused into the extraction kernel (mfcc_forward_sums)" if args.fused_sums and not args.overlap else
                          ("per chunk on a second stream" if args.overlap else "one pass over the resident rows (mfcc_cmvn_accumulate)"),
            "checks": dict(checks, frames_in_sums=n_in_sums),
        }
        print(json.dumps(rec), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--variant", default="self5", choices=("self5", "self15"))
    ap.add_argument("--config", default="B", choices=("B", "D"),
                    help="B: 10 000 x 3 s -> MFCC-39 per GPU (headline); D: --hours of audio sharded over the GPUs + corpus CMVN")
    ap.add_argument("--hours", type=float, default=None, help="config D: corpus size in hours (default 125 per GPU: 1000 h on 8)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the compact config C / E figures")
    ap.add_argument("--backend", default="nccl", choices=("nccl", "gloo"),
                    help="process-group backend for N > 1: nccl = RCCL over xGMI (default); gloo lets N ranks share one GPU (tests)")
    ap.add_argument("--two-pass-sums", dest="fused_sums", action="store_false",
                    help="config D: collect the statistics in a pass of their own over the resident rows (mfcc_cmvn_accumulate) "
                         "instead of inside the extraction kernel (mfcc_forward_sums, the default: measured 0.4 ms of 22 faster)")
    ap.add_argument("--overlap", action="store_true",
                    help="config D: per-chunk statistics passes on a second stream beside the next chunk's extraction, instead of "
                         "one pass after the last chunk (measured slower: the two kernels share the CUs, profiles/r3_configD_*.json)")
    ap.add_argument("--no-precondition", action="store_true", help="skip the untimed clock-ramp launches in front of the warm-up")
    args = ap.parse_args()
    # Defaults of config B: the part ramps its clocks up over the first ~35 launches of a cold start
    # (per-launch times in profiles/r2_final_kernel_trace_durations.txt), so 60 untimed launches come first.
    if args.steps is None:
        args.steps = 200 if args.config == "B" else 3
    if args.warmup is None:
        args.warmup = 60 if args.config == "B" else 1

    if args.gpus > 1 and "RANK" not in os.environ:
        # started by hand without a launcher: start one process per GPU and relay the result
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))

    cpu = None
    if world == 1 and not args.no_cpu_baseline and args.config == "B":
        cpu = cpu_baseline()          # before the GPU is touched: the pool must not inherit a HIP context

    import torch
    import mfcc_amd

    # one GPU per rank; under gloo several ranks may share a GPU (a one-GPU box rehearsing the N-rank path)
    dev_index = local_rank if args.backend == "nccl" else local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")

    if args.config == "D":
        if args.hours is None:
            args.hours = 125.0 * world
        run_config_d(args, rank, world, dev, torch, mfcc_amd, dist)
        if world > 1:
            dist.destroy_process_group()
        return

    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    audio = synth_audio(N_UTT, dev, gen, torch)
    fe = mfcc_amd.MFCC(args.variant, device=dev)
    layout = fe.layout([UTT_LEN] * N_UTT)
    frames = layout.total_frames
    assert frames == N_UTT * FRAMES_PER_UTT
    out = torch.empty((frames, 39), dtype=torch.float32, device=dev)
    flat = audio.view(-1)

    def step():
        fe.forward(flat, layout, out=out, kind="mfcc39")

    pre = None if args.no_precondition else precondition(step, torch)
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
        elapsed = max_over_ranks(elapsed, dist, torch, dev)
        # the one exchange of the front-end, outside the timed region: corpus CMVN sums
        sums = mfcc_amd.cmvn_allreduce(mfcc_amd.cmvn_accumulate(out))
        assert int(sums[0].item()) == frames * world

    assert torch.isfinite(out[:: 9973]).all()
    extra = None
    if rank == 0 and world == 1 and not args.no_extra:
        del out
        torch.cuda.empty_cache()
        extra = extra_measurements(torch, mfcc_amd, dev, audio, fe, layout)
    if rank == 0:
        total_frames = frames * world
        value = total_frames * args.steps / elapsed
        bytes_per_launch = 2 * N_UTT * UTT_LEN + 4 * 39 * frames          # int16 in + fp32 out
        achieved = bytes_per_launch / (kernel_ms * 1e-3) / 1e9
        traffic = traffic_bytes = traffic_source = None
        rev = kernel_revision()
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tpath) and args.variant == "self5":
            # HBM bytes per launch of this very workload, from separate rocprofv3 --pmc passes
            # (FETCH_SIZE doubled as gfx950 requires, + WRITE_SIZE); counters cannot run inside this
            # process.  Used only while the file belongs to the kernel sources that are running.
            with open(tpath) as fh:
                tj = json.load(fh)
            if tj.get("kernel_revision") == rev:
                traffic_bytes = tj["hbm_bytes_per_launch"]
                traffic = round(traffic_bytes / (kernel_ms * 1e-3) / 1e9, 1)
                traffic_source = f"profiles/hbm_traffic.json (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, kernel sources {rev})"
            else:
                traffic_source = f"none: profiles/hbm_traffic.json was measured on kernel sources {tj.get('kernel_revision')}, running {rev}"
        flops_per_frame = 10600.0   # split-radix real FFT-512 7.7 k + |X|^2 0.8 k + sparse mel 0.9 k + DCT 0.7 k + ln, delta 0.5 k
        fp32_tflops = flops_per_frame * frames / (kernel_ms * 1e-3) / 1e12
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
                "traffic_source": traffic_source,
                "kernel_revision": rev,
            },
            "roofline_fp32": {
                "bound": "valu", "achieved": round(fp32_tflops, 2), "peak": 157.3, "unit": "TFLOP/s",
                "frac": round(fp32_tflops / 157.3, 4), "flops_per_frame": flops_per_frame,
                "note": "minimum-work count (split-radix real FFT, sparse filterbank); secondary line, the HBM roofline is the headline",
            },
        }
        if pre is not None:
            rec["preconditioning"] = pre
        if cpu is not None:
            rec["cpu_baseline"] = cpu
        if extra is not None:
            rec["extra"] = extra
        print(json.dumps(rec), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
