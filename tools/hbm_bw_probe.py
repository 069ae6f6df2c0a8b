import torch, time
x = torch.empty(13_000_000_000, dtype=torch.float32, device="cuda")  # 52 GB
def t(fn, n=3):
    fn(); torch.cuda.synchronize()
    a=torch.cuda.Event(enable_timing=True); b=torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b)/n
ms = t(lambda: x.zero_())
print(f"zero_ 52 GB: {ms:.2f} ms -> {52e9/ms/1e6:.0f} GB/s")
ms = t(lambda: x.fill_(1.5))
print(f"fill_ 52 GB: {ms:.2f} ms -> {52e9/ms/1e6:.0f} GB/s")
y = x[:6_500_000_000]; z = x[6_500_000_000:]
ms = t(lambda: z.copy_(y))
print(f"copy 26 GB -> 26 GB: {ms:.2f} ms -> {52e9/ms/1e6:.0f} GB/s (r+w)")
