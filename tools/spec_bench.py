#!/usr/bin/env python3
"""257-bin spectrum rows of the bench batch: mfcc_forward(kind=spec/logspec) against the row writer of
mfcc_forward_stacked with a window of one frame (same values, different store pattern)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mfcc_amd
import mfcc_amd.frontend as F
U = 10000
audio = (torch.randn(U, 48000, device="cuda") * 3000).clamp(-32768, 32767).to(torch.int16).view(-1)
fe = mfcc_amd.MFCC("self5"); lay = fe.layout([48000] * U)
out = torch.empty((lay.total_frames, 257), dtype=torch.float32, device="cuda")
def timed(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
alg = 2 * U * 48000 + 1028 * lay.total_frames
for log in (True, False):
    t1 = timed(lambda: fe.forward(audio, lay, out=out, kind="logspec" if log else "spec"))
    ref = out.clone()
    t2 = timed(lambda: F.spectrum_stack(fe, audio, lay, 1, out=out, log=log))
    print(f"log={log}: forward kind {t1:.3f} ms ({alg / t1 / 1e6:.0f} GB/s), stacked N=1 {t2:.3f} ms ({alg / t2 / 1e6:.0f} GB/s), equal {torch.equal(ref, out)}")
wide = torch.empty((lay.total_frames, 260), dtype=torch.float32, device="cuda")
t3 = timed(lambda: fe.forward(audio, lay, out=wide[:, :257], kind="spec"))
print(f"row pitch 260 (register path): {t3:.3f} ms")
st = torch.empty((lay.total_frames, 13), dtype=torch.float32, device="cuda")
t4 = timed(lambda: fe.forward(audio, lay, out=st, kind="static"))
print(f"static kind: {t4:.3f} ms")
