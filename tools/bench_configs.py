#!/usr/bin/env python3
"""Secondary measurements (one JSON line each), run on one MI355X:

  C    257-bin log power spectrum + 17-frame context stacking, 10 000 x 3 s utterances
  D    one GPU's share of the 1000 h corpus: 150 000 x 3 s utterances -> MFCC-39 kept in HBM, fp64 CMVN sums,
       constants, in-place normalisation (the all-reduce of 79 doubles is a no-op at world size 1)
  E    streaming: B = 1..64 lock-step streams x 10 ms hops, host-observed latency per hop
  pcie numpy in -> numpy out through mfcc39() (host buffers cross the link; never the headline value)

    python tools/bench_configs.py [--only C,D,E,pcie] [--utts-d 150000]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import mfcc_amd  # noqa: E402

UTT_LEN, FPU = 48000, 299
HBM = 8000.0


def synth(n, gen, out=None):
    if out is None:
        out = torch.empty((n, UTT_LEN), dtype=torch.int16, device="cuda")
    for i in range(0, n, 1000):
        m = min(1000, n - i)
        out[i:i + m] = (torch.randn(m, UTT_LEN, generator=gen, device="cuda") * 3000).clamp(-32768, 32767).to(torch.int16)
    return out


def timed(fn, steps, warmup=2):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in ev)
    return ms[0], ms[len(ms) // 2]


def config_b_variants(args):
    """The headline shape through the other entry conditions: magnitude variant, float32 samples, static-only."""
    n = 10000
    gen = torch.Generator(device="cuda").manual_seed(1234)
    audio = synth(n, gen)
    faudio = (audio.float() / 32768).view(-1)
    for variant, buf, kind in (("self5", audio.view(-1), "mfcc39"), ("self15", audio.view(-1), "mfcc39"),
                               ("self5", faudio, "mfcc39"), ("self5", audio.view(-1), "static"),
                               ("self5", audio.view(-1), "logmel")):
        fe = mfcc_amd.MFCC(variant)
        lay = fe.layout([UTT_LEN] * n)
        out = torch.empty((lay.total_frames, fe.plan.out_width(kind)), dtype=torch.float32, device="cuda")
        t = timed(lambda: fe.forward(buf, lay, out=out, kind=kind), 20, 3)
        alg = buf.element_size() * buf.numel() + 4 * out.shape[1] * lay.total_frames
        print(json.dumps({"config": f"B variant: {variant}, {str(buf.dtype).split('.')[-1]} samples, kind {kind}",
                          "ms_min_med": [round(v, 4) for v in t], "frames_per_s": round(lay.total_frames / (t[1] * 1e-3), 1),
                          "algorithmic_bytes": alg, "roofline_frac": round(alg / (t[1] * 1e-3) / 1e9 / HBM, 4)}), flush=True)


def config_general(args):
    """The general-geometry kernel on the headline shape (forced) and at 8 kHz with the derived 256-point FFT."""
    n = 10000
    gen = torch.Generator(device="cuda").manual_seed(1234)
    audio = synth(n, gen).view(-1)
    for name, kw, fs in (("general kernel forced, 400/160/512", dict(kernel="general"), 16000),
                         ("fs=8000 -> 200/80 samples, 256-point FFT (specialised kernel, even bins of its 512-point transform), 6 s utterances", dict(fs=8000), 8000),
                         ("general kernel forced, fs=8000 -> 200/80/256", dict(fs=8000, kernel="general"), 8000),
                         ("general kernel, pad_fac=3", dict(pad_fac=3), 16000),
                         ("frame_time=20 -> 320/160/512 (specialised kernel, third geometry)", dict(frame_time=20), 16000),
                         ("general kernel forced, frame_time=20 -> 320/160/512", dict(frame_time=20, kernel="general"), 16000),
                         ("general kernel, fs=44100 -> 1102/441/2048 (64 lanes per frame)", dict(fs=44100), 44100),
                         ("general kernel, fs=22050 -> 551/220/1024 (32 lanes per frame)", dict(fs=22050), 22050)):
        fe = mfcc_amd.MFCC("self5", **kw)
        lay = fe.layout([UTT_LEN] * n)
        out = torch.empty((lay.total_frames, 39), dtype=torch.float32, device="cuda")
        t = timed(lambda: fe.forward(audio, lay, out=out), 5, 1)
        alg = 2 * audio.numel() + 156 * lay.total_frames
        print(json.dumps({"config": name, "general": fe.plan.general, "frames": lay.total_frames, "ms_min_med": [round(v, 4) for v in t],
                          "frames_per_s": round(lay.total_frames / (t[1] * 1e-3), 1),
                          "roofline_frac": round(alg / (t[1] * 1e-3) / 1e9 / HBM, 4)}), flush=True)


def config_c(args):
    n = 10000
    gen = torch.Generator(device="cuda").manual_seed(1234)
    audio = synth(n, gen).view(-1)
    fe = mfcc_amd.MFCC("self5")
    lay = fe.layout([UTT_LEN] * n)
    frames = lay.total_frames
    spec = torch.empty((frames, 257), dtype=torch.float32, device="cuda")
    ctx = torch.empty((frames, 17 * 257), dtype=torch.float32, device="cuda")
    t_spec = timed(lambda: fe.forward(audio, lay, out=spec, kind="logspec"), 10)
    t_stack = timed(lambda: mfcc_amd.stack_context(spec, 17, frame_counts=[FPU] * n, out=ctx), 5)

    def both():
        fe.forward(audio, lay, out=spec, kind="logspec")
        mfcc_amd.stack_context(spec, 17, frame_counts=[FPU] * n, out=ctx)
    t_both = timed(both, 5)
    del spec
    torch.cuda.empty_cache()
    t_one = timed(lambda: mfcc_amd.spectrum_stack(fe, audio, lay, 17, out=ctx), 5)   # mfcc_forward_stacked: one launch
    alg = 2 * n * UTT_LEN + 4 * 257 * 17 * frames
    rec = {"config": "C: 257-bin log power spectrum + 17-frame stack, 10000 x 3 s", "frames": frames,
           "spec_ms_min_med": t_spec, "stack_ms_min_med": t_stack, "total_ms_min_med": t_both,
           "frames_per_s": round(frames / (t_both[1] * 1e-3), 1), "algorithmic_bytes": alg,
           "achieved_GBps": round(alg / (t_both[1] * 1e-3) / 1e9, 1),
           "roofline_frac": round(alg / (t_both[1] * 1e-3) / 1e9 / HBM, 4),
           "one_launch_ms_min_med": t_one, "one_launch_frames_per_s": round(frames / (t_one[1] * 1e-3), 1),
           "one_launch_roofline_frac": round(alg / (t_one[1] * 1e-3) / 1e9 / HBM, 4),
           "spec_only_bytes": 2 * n * UTT_LEN + 1028 * frames,
           "spec_only_frac": round((2 * n * UTT_LEN + 1028 * frames) / (t_spec[1] * 1e-3) / 1e9 / HBM, 4)}
    print(json.dumps(rec), flush=True)


def config_d(args):
    n_total, chunk = args.utts_d, 10000
    gen = torch.Generator(device="cuda").manual_seed(1234)
    fe = mfcc_amd.MFCC("self5")
    lay = fe.layout([UTT_LEN] * chunk)
    feats = torch.empty((n_total * FPU, 39), dtype=torch.float32, device="cuda")
    audio = torch.empty((chunk, UTT_LEN), dtype=torch.int16, device="cuda")
    sums = torch.zeros(79, dtype=torch.float64, device="cuda")
    t_fwd = t_acc = 0.0
    a, b, c = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    for i in range(0, n_total, chunk):
        synth(chunk, gen, audio)            # regenerated per chunk, outside the clock
        view = feats[i * FPU:(i + chunk) * FPU]
        a.record()
        fe.forward(audio.view(-1), lay, out=view)
        b.record()
        mfcc_amd.cmvn_accumulate(view, sums)
        c.record()
        torch.cuda.synchronize()
        t_fwd += a.elapsed_time(b)
        t_acc += b.elapsed_time(c)
    mfcc_amd.cmvn_constants(torch.ones(79, dtype=torch.float64, device="cuda") * 4)  # loads torch's small kernels
    torch.cuda.synchronize()
    a.record()
    sums = mfcc_amd.cmvn_allreduce(sums)
    mm, vv = mfcc_amd.cmvn_constants(sums)
    b.record()
    mfcc_amd.cmvn_apply(feats, mm, vv)
    c.record()
    torch.cuda.synchronize()
    t_const, t_apply = a.elapsed_time(b), b.elapsed_time(c)
    # check against torch in float64 on a slice
    chk = feats[: 50 * FPU].double()
    frames = n_total * FPU
    total_ms = t_fwd + t_acc + t_const + t_apply
    alg = 2 * n_total * UTT_LEN + 468 * frames
    rec = {"config": f"D (one GPU's share): {n_total} x 3 s -> MFCC-39 + corpus CMVN", "frames": frames,
           "audio_hours": round(n_total * 3 / 3600, 1), "forward_ms": round(t_fwd, 2), "accumulate_ms": round(t_acc, 2),
           "constants_ms": round(t_const, 3), "apply_ms": round(t_apply, 2), "total_ms": round(total_ms, 2),
           "frames_per_s": round(frames / (total_ms * 1e-3), 1), "algorithmic_bytes": alg,
           "roofline_frac": round(alg / (total_ms * 1e-3) / 1e9 / HBM, 4),
           "n_in_sums": int(sums[0].item()), "normalised_mean_abs_max": float(feats.double().mean(0).abs().max().item()),
           "normalised_std_err_max": float((feats.double().std(0, unbiased=False) - 1).abs().max().item()),
           "finite_check_slice": bool(torch.isfinite(chk).all().item())}
    print(json.dumps(rec), flush=True)


def config_e(args):
    hops = 1000
    for B in (1, 2, 4, 8, 16, 32, 64):
        s = mfcc_amd.StreamingMFCC("self5", n_streams=B, max_chunk=160)
        audio = (torch.randn(B, 160 * hops, device="cuda") * 3000).clamp(-32768, 32767).to(torch.int16)
        out = torch.empty((B, s.max_rows, 39), dtype=torch.float32, device="cuda")
        for h in range(20):
            s.push(audio[:, h * 160:(h + 1) * 160], out=out)
        torch.cuda.synchronize()
        lat = []
        rows = 0
        t_all = time.perf_counter()
        for h in range(20, hops):
            t0 = time.perf_counter()
            r = s.push(audio[:, h * 160:(h + 1) * 160], out=out)
            torch.cuda.synchronize()
            lat.append((time.perf_counter() - t0) * 1e6)
            rows += r.shape[1]
        wall = time.perf_counter() - t_all
        s.flush(out=out)
        lat.sort()
        # the same hops without a host wait per hop: launch-rate bound
        s2 = mfcc_amd.StreamingMFCC("self5", n_streams=B, max_chunk=160)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for h in range(hops):
            s2.push(audio[:, h * 160:(h + 1) * 160], out=out)
        torch.cuda.synchronize()
        wall2 = time.perf_counter() - t0
        rec = {"config": "E: streaming 10 ms hops", "streams": B, "hops": hops - 20,
               "us_per_hop_p50": round(lat[len(lat) // 2], 1), "us_per_hop_p99": round(lat[int(len(lat) * 0.99)], 1),
               "frames_per_s_sync_each_hop": round(B * rows / wall, 1),
               "us_per_hop_async": round(wall2 / hops * 1e6, 1), "frames_per_s_async": round(B * hops / wall2, 1),
               "x_realtime_per_stream_async": round(0.010 / (wall2 / hops), 1)}
        print(json.dumps(rec), flush=True)


def config_pcie(args):
    n = 2000
    rng = np.random.default_rng(0)
    audio = np.clip(rng.standard_normal((n, UTT_LEN)) * 3000, -32768, 32767).astype(np.int16)
    mfcc_amd.mfcc39(audio[:100])
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        out = mfcc_amd.mfcc39(audio)
    dt = (time.perf_counter() - t0) / reps
    rec = {"config": "PCIe-inclusive: numpy int16 [2000, 48000] in -> numpy float32 [598000, 39] out (pageable host memory)",
           "ms": round(dt * 1e3, 2), "frames_per_s": round(out.shape[0] / dt, 1),
           "host_bytes_moved": audio.nbytes + out.nbytes, "host_link_GBps": round((audio.nbytes + out.nbytes) / dt / 1e9, 2)}
    print(json.dumps(rec), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="B,G,C,D,E,pcie")
    ap.add_argument("--utts-d", type=int, default=150000)
    args = ap.parse_args()
    which = set(args.only.split(","))
    if "B" in which:
        config_b_variants(args)
        torch.cuda.empty_cache()
    if "G" in which:
        config_general(args)
        torch.cuda.empty_cache()
    if "C" in which:
        config_c(args)
        torch.cuda.empty_cache()
    if "D" in which:
        config_d(args)
        torch.cuda.empty_cache()
    if "E" in which:
        config_e(args)
    if "pcie" in which:
        config_pcie(args)


if __name__ == "__main__":
    main()
