#!/usr/bin/env python3
"""Do two builds of the library give the same bits?  MFCC-39 rows of 50 seeded 3 s utterances through each.

    python3 tools/lib_equal.py mfcc_amd/lib/libmfcc_amd.so build/variants/libmfcc_x.so
"""
import sys, os, torch
sys.path.insert(0, '/root/repo' if os.path.exists('/root/repo/mfcc_amd') else '.')
from mfcc_amd import _capi as capi
import mfcc_amd.frontend as F
outs=[]
gen = torch.Generator(device="cuda").manual_seed(7)
audio=(torch.randn(50,48000,generator=gen,device="cuda")*3000).clamp(-32768,32767).to(torch.int16).view(-1)
for path in sys.argv[1:]:
    capi._lib=None; capi.LIB_PATH=os.path.abspath(path); F._plan_cache.clear()
    fe=F.MFCC("self5"); lay=fe.layout([48000]*50)
    out=torch.empty((lay.total_frames,39),dtype=torch.float32,device="cuda")
    fe.forward(audio,lay,out=out,kind="mfcc39"); torch.cuda.synchronize()
    outs.append(out.clone())
print("bit-equal:", torch.equal(outs[0],outs[1]), "max diff", (outs[0]-outs[1]).abs().max().item())
