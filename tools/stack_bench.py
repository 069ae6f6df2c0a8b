import sys, os, torch
sys.path.insert(0, os.getcwd())
from mfcc_amd import _capi as capi
lib = sys.argv[1]
capi.LIB_PATH = os.path.abspath(lib)
import mfcc_amd
n=10000; F=299
spec = torch.randn(n*F, 257, device="cuda")
ctx = torch.empty(n*F, 17*257, device="cuda")
fc=[F]*n
def run(): mfcc_amd.stack_context(spec, 17, frame_counts=fc, out=ctx)
run(); torch.cuda.synchronize()
a=torch.cuda.Event(enable_timing=True); b=torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(5): run()
b.record(); torch.cuda.synchronize()
ms=a.elapsed_time(b)/5
print(f"{lib}: stack {ms:.2f} ms  {ctx.numel()*4/ms/1e6:.0f} GB/s written")
