#!/usr/bin/env python3
"""Diagnostic build only (make variant NAME=splitdiag EXTRA="-DMFCC_DIAG -DMFCC_USE_SPLIT"): where the two roles of
split_kernel.h spend a tile's time -- work before and after barrier X and the waits at both barriers, from s_memtime
stamps summed per wave."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from mfcc_amd import _capi as capi  # noqa: E402
import mfcc_amd.frontend as F  # noqa: E402

capi.LIB_PATH = os.path.join(ROOT, "build", "variants", "libmfcc_splitdiag.so")
lib = capi.load()
utts = 10000
audio = (torch.randn(utts, 48000, device="cuda") * 3000).clamp(-32768, 32767).to(torch.int16)
fe = F.MFCC("self5")
lay = fe.layout([48000] * utts)
out = torch.empty((lay.total_frames, 39), dtype=torch.float32, device="cuda")
dbg = torch.zeros((utts, 8, 8), dtype=torch.int64, device="cuda")
fe.forward(audio.view(-1), lay, out=out)
torch.cuda.synchronize()
lib.mfcc_debug_set_buffer.argtypes = [C.c_void_p]
lib.mfcc_debug_set_buffer(C.c_void_p(dbg.data_ptr()))
fe.forward(audio.view(-1), lay, out=out)
torch.cuda.synchronize()
d = dbg.double()
d = d[d[:, 0, 7] > 0]          # workgroups that ran (the grid is 2 x CUs persistent workgroups)
print("workgroups:", d.shape[0], " steps per workgroup (mean, min, max):", d[:, 0, 4].mean().item(), d[:, 0, 4].min().item(), d[:, 0, 4].max().item())
names = ["work before X", "wait at X", "work after X", "wait at Y"]
for role, sl in (("T (waves 0-3)", slice(0, 4)), ("F (waves 4-7)", slice(4, 8))):
    m = d[:, sl, :].mean(dim=(0, 1))
    tot = m[7].item()
    print(role, f"kernel lifetime {tot:.0f} ticks")
    if role.startswith("F"):
        print(f"   of 'before X': filterbank + DCT {m[5].item() / m[4].item():8.0f} per step;  of 'after X': staging store {m[6].item() / m[4].item():8.0f} per step")
    for i, n in enumerate(names):
        print(f"   {n:16s} {m[i].item():10.0f} ticks  {m[i].item() / tot * 100:6.2f} %   per step {m[i].item() / m[4].item():8.0f}")
