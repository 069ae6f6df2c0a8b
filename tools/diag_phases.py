#!/usr/bin/env python3
"""Diagnostic build only (make variant NAME=diag EXTRA=-DMFCC_DIAG): per-phase cycle shares of
the fused kernel's batch loop, from s_memtime stamps summed per wave."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from mfcc_amd import _capi as capi  # noqa: E402
import mfcc_amd.frontend as F  # noqa: E402

capi.LIB_PATH = os.path.join(ROOT, "build", "variants", "libmfcc_diag.so")
lib = capi.load()
utts = 10000
audio = (torch.randn(utts, 48000, device="cuda") * 3000).clamp(-32768, 32767).to(torch.int16)
fe = F.MFCC("self5")
lay = fe.layout([48000] * utts)
out = torch.empty((lay.total_frames, 39), dtype=torch.float32, device="cuda")
dbg = torch.zeros((utts, 4, 8), dtype=torch.int64, device="cuda")
fe.forward(audio.view(-1), lay, out=out)
torch.cuda.synchronize()
lib.mfcc_debug_set_buffer.argtypes = [C.c_void_p]
lib.mfcc_debug_set_buffer(C.c_void_p(dbg.data_ptr()))
fe.forward(audio.view(-1), lay, out=out)
torch.cuda.synchronize()
d = dbg.double()
names = ["fft", "wait_spec_barrier", "mel+ln", "staging", "wait_barrier2", "dct(duty)", "loop_total", "kernel_total"]
m = d.mean(dim=(0, 1))
for n, v in zip(names, m.tolist()):
    print(f"{n:20s} {v:12.0f} ticks/wave   {v / m[7].item() * 100:6.2f} % of wave lifetime   per batch {v / 19:9.0f}")
print("per-wave (wave 0..3) loop totals:", d[:, :, 6].mean(dim=0).tolist())
