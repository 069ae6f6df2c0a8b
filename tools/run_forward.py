#!/usr/bin/env python3
"""Run the fused MFCC-39 kernel a few times on the benchmark batch (for rocprofv3 passes and
quick A/B timing of library builds).

    python3 tools/run_forward.py [--steps 5] [--utts 10000] [--kind mfcc39] [--lib path.so ...]

With several --lib arguments every library is timed in the same process, interleaved."""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--utts", type=int, default=10000)
    ap.add_argument("--kind", default="mfcc39")
    ap.add_argument("--variant", default="self5")
    ap.add_argument("--lib", action="append", default=[])
    args = ap.parse_args()
    import torch
    from mfcc_amd import _capi as capi
    import mfcc_amd.frontend as F

    libs = args.lib or [capi.LIB_PATH]
    gen = torch.Generator(device="cuda").manual_seed(1234)
    audio = torch.empty((args.utts, 48000), dtype=torch.int16, device="cuda")
    for i in range(0, args.utts, 1000):
        n = min(1000, args.utts - i)
        audio[i:i + n] = (torch.randn(n, 48000, generator=gen, device="cuda") * 3000).clamp(-32768, 32767).to(torch.int16)
    flat = audio.view(-1)
    results = {}
    ctx = []
    for path in libs:
        capi._lib = None
        capi.LIB_PATH = os.path.abspath(path)
        F._plan_cache.clear()
        fe = F.MFCC(args.variant)
        lay = fe.layout([48000] * args.utts)
        out = torch.empty((lay.total_frames, fe.plan.out_width(args.kind)), dtype=torch.float32, device="cuda")
        fe.forward(flat, lay, out=out, kind=args.kind)
        torch.cuda.synchronize()
        ctx.append((path, fe, lay, out))
        results[path] = []
    for _ in range(args.rounds):
        for path, fe, lay, out in ctx:
            capi._lib = None
            capi.LIB_PATH = os.path.abspath(path)
            capi.load()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(args.steps):
                fe.forward(flat, lay, out=out, kind=args.kind)
            e.record()
            torch.cuda.synchronize()
            results[path].append(s.elapsed_time(e) / args.steps)
    ref = None
    for path, fe, lay, out in ctx:
        ms = results[path]
        chk = float(out[::997].double().sum())
        ref = chk if ref is None else ref
        print(f"{os.path.basename(path):40s} min {min(ms):8.4f} ms  med {sorted(ms)[len(ms)//2]:8.4f} ms  "
              f"{lay.total_frames / min(ms) / 1e6:8.2f} Gframes/s  checksum {chk:.6e}", flush=True)


if __name__ == "__main__":
    main()
