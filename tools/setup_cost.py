#!/usr/bin/env python3
"""What a tile costs beyond its batches: the MFCC-39 kernel on batches of equal total length cut into utterances of
3 s, 1.5 s, 6 s and 0.75 s (19, 10, 19 + halo and 5 batches per tile).  T = tiles * S + batches * B gives S and B.

    python3 tools/setup_cost.py
"""
import sys, os, torch
sys.path.insert(0, '.')
import mfcc_amd.frontend as F
fe = F.MFCC("self5")
gen = torch.Generator(device="cuda").manual_seed(1)
res = {}
for n_utt, L in ((10000, 48000), (20000, 24000-80), (5000, 96000+400-160*0), (40000, 12000-80*2)):
    audio = (torch.randn(n_utt * L, generator=gen, device="cuda") * 3000).clamp(-32768, 32767).to(torch.int16)
    lay = fe.layout([L] * n_utt)
    out = torch.empty((lay.total_frames, 39), dtype=torch.float32, device="cuda")
    for _ in range(60): fe.forward(audio, lay, out=out)
    torch.cuda.synchronize()
    best = 1e9
    for r in range(5):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(30): fe.forward(audio, lay, out=out)
        e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / 30)
    nf = lay.total_frames // n_utt
    print(f"{n_utt} x {L}: frames/utt {nf}, batches/tile {(nf+15)//16 if nf <= 304 else '?'}, {best:.4f} ms, {best*1e6/lay.total_frames:.4f} ns/frame, per 16-frame slot {best*1e6/(n_utt*((nf+15)//16)):.2f} ns")
    del audio, out
