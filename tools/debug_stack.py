import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import mfcc_amd as M
rng = np.random.default_rng(33)
frames = [1, 2, 3, 8, 9, 16, 17, 40, 299, 304, 305, 700]
lengths = [400 + 160 * (f - 1) - int(rng.integers(0, 160)) if f > 1 else 123 for f in frames]
audios = [np.clip(rng.standard_normal(n) * 3000, -32768, 32767).astype(np.int16) for n in lengths]
fe = M.MFCC("self5")
lay = fe.layout(lengths)
packed = torch.from_numpy(np.concatenate(audios)).cuda()
spec = fe.forward(packed, lay, kind="logspec")
counts = np.diff(lay.frame_offsets)
for N in (17, 5, 1):
    two = M.stack_context(spec, N, frame_counts=counts).cpu().numpy()
    out = torch.full((sum(frames), N * 257), 7777.0, device="cuda")
    one = M.spectrum_stack(fe, packed, lay, N, out=out, log=True).cpu().numpy()
    bad = np.argwhere(one != two)
    print("N", N, "mismatches", len(bad), "of", one.size)
    fo = lay.frame_offsets
    seen = 0
    for u in range(len(frames)):
        b = bad[(bad[:, 0] >= fo[u]) & (bad[:, 0] < fo[u + 1])]
        if len(b):
            rows = sorted(set((b[:, 0] - fo[u]).tolist()))
            print("  utt", u, "frames", frames[u], "bad rows", rows[:20], "n", len(b))
            r, c = b[0]
            print("    first", r - fo[u], "col", c, "block", c // 257, "c", c % 257, "one", one[r, c], "two", two[r, c])
            cols = sorted(set(b[b[:, 0] == r][:, 1].tolist()))
            print("    cols in that row", cols[:12], "...", cols[-6:], len(cols))
