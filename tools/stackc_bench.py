#!/usr/bin/env python3
"""A/B timing of the one-launch stacked-spectrum path (config C shape) for several library builds.
    python3 tools/stackc_bench.py --lib a.so --lib b.so"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser(); ap.add_argument("--lib", action="append", default=[]); ap.add_argument("--utts", type=int, default=10000)
args = ap.parse_args()
import torch
from mfcc_amd import _capi as capi
import mfcc_amd.frontend as F
audio = (torch.randn(args.utts, 48000, device="cuda") * 3000).clamp(-32768, 32767).to(torch.int16).view(-1)
ctx = torch.empty((args.utts * 299, 17 * 257), dtype=torch.float32, device="cuda")
for path in args.lib or [capi.LIB_PATH]:
    capi._lib = None; capi.LIB_PATH = os.path.abspath(path); F._plan_cache.clear()
    fe = F.MFCC("self5"); lay = fe.layout([48000] * args.utts)
    for _ in range(2): F.spectrum_stack(fe, audio, lay, 17, out=ctx)
    torch.cuda.synchronize()
    ms = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); F.spectrum_stack(fe, audio, lay, 17, out=ctx); b.record(); torch.cuda.synchronize(); ms.append(a.elapsed_time(b))
    alg = 2 * args.utts * 48000 + 4 * 257 * 17 * lay.total_frames
    print(f"{os.path.basename(path):32s} min {min(ms):7.3f} ms med {sorted(ms)[2]:7.3f} ms  {alg / min(ms) / 1e6:7.1f} GB/s  frac {alg / min(ms) / 1e6 / 8000:.3f}  chk {float(ctx[::99991].double().sum()):.6e}", flush=True)
