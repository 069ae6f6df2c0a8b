#!/usr/bin/env python3
"""MFCC-39-style rows for a numcep other than 13 (run-time coefficient count) on the bench batch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, mfcc_amd
U = 10000
audio = (torch.randn(U, 48000, device="cuda") * 3000).clamp(-32768, 32767).to(torch.int16).view(-1)
for nc in (13, 12, 16):
    fe = mfcc_amd.MFCC("self5", numcep=nc); lay = fe.layout([48000] * U)
    out = torch.empty((lay.total_frames, 3 * nc), dtype=torch.float32, device="cuda")
    for _ in range(20): fe.forward(audio, lay, out=out)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(40): fe.forward(audio, lay, out=out)
    b.record(); torch.cuda.synchronize()
    print(f"numcep {nc}: {a.elapsed_time(b) / 40:.3f} ms", flush=True)
