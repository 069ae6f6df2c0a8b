#!/usr/bin/env python3
"""Fused MFCC-39 kernel on data that is not in any cache: K different 10 000-utterance chunks (audio and
output regions of their own), one launch each, round robin -- the access pattern of the corpus
pipeline.  Compare with tools/run_forward.py, which reuses one chunk.
    python3 tools/cold_forward.py [--chunks 15] [--lib a.so --lib b.so]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--chunks", type=int, default=15)
ap.add_argument("--lib", action="append", default=[])
args = ap.parse_args()
import torch
from mfcc_amd import _capi as capi
import mfcc_amd.frontend as F
U, L = 10000, 48000
audio = torch.empty((args.chunks, U * L), dtype=torch.int16, device="cuda")
for c in range(args.chunks):
    for i in range(0, U, 2000):
        audio[c, i * L:(i + 2000) * L] = (torch.randn(2000 * L, device="cuda") * 3000).clamp(-32768, 32767).to(torch.int16)
out = torch.empty((args.chunks, U * 299, 39), dtype=torch.float32, device="cuda")
for path in args.lib or [capi.LIB_PATH]:
    capi._lib = None; capi.LIB_PATH = os.path.abspath(path); F._plan_cache.clear()
    fe = F.MFCC("self5"); lay = fe.layout([L] * U)
    for rep in range(4):
        for c in range(args.chunks):
            fe.forward(audio[c], lay, out=out[c])
    torch.cuda.synchronize()
    res = []
    for rep in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for c in range(args.chunks):
            fe.forward(audio[c], lay, out=out[c])
        b.record(); torch.cuda.synchronize()
        res.append(a.elapsed_time(b) / args.chunks)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for c in range(45):
        fe.forward(audio[0], lay, out=out[0])
    b.record(); torch.cuda.synchronize()
    print(f"{os.path.basename(path):32s} cold chunks: min {min(res):.4f} ms/launch  same chunk: {a.elapsed_time(b) / 45:.4f} ms/launch", flush=True)
