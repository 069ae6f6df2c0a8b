#!/usr/bin/env python3
"""Times of the stacked kinds on the bench batch (10 000 x 3 s): MFCC-39 x N in one launch vs forward +
stack_context, and the one-launch spectrum stack for window lengths of both 16-byte phases and both ring sizes."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import mfcc_amd  # noqa: E402


def timed(fn, n=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    return sorted(a.elapsed_time(b) for a, b in ev)[n // 2]


def main():
    n_utt, n_len = 10000, 48000
    audio = torch.empty(n_utt * n_len, dtype=torch.int16, device="cuda")
    for i in range(0, n_utt, 1000):
        audio[i * n_len:(i + 1000) * n_len] = (torch.randn(1000 * n_len, device="cuda") * 3000).clamp(-32768, 32767).to(torch.int16)
    fe = mfcc_amd.MFCC("self5")
    lay = fe.layout([n_len] * n_utt)
    frames = lay.total_frames
    rows = torch.empty((frames, 39), dtype=torch.float32, device="cuda")
    for _ in range(40):
        fe.forward(audio, lay, out=rows)
    print(f"mfcc39 plain                     {timed(lambda: fe.forward(audio, lay, out=rows), 9):8.3f} ms")
    for n in (11, 21):
        out = torch.empty((frames, n * 39), dtype=torch.float32, device="cuda")
        one = timed(lambda: mfcc_amd.mfcc39_stack(fe, audio, lay, n, out=out))
        two = timed(lambda: mfcc_amd.stack_context(fe.forward(audio, lay, out=rows), n, frame_counts=[299] * n_utt, out=out))
        alg = 2 * n_utt * n_len + 4 * 39 * n * frames
        print(f"mfcc39 x {n:2d}: one launch {one:8.3f} ms ({alg / one / 1e6 / 8000:.3f} of the HBM roofline on {alg / 1e9:.2f} GB)   forward + stack_context {two:8.3f} ms")
        del out
    for n in (9, 11, 17, 21):
        out = torch.empty((frames, n * 257), dtype=torch.float32, device="cuda")
        one = timed(lambda: mfcc_amd.spectrum_stack(fe, audio, lay, n, out=out))
        alg = 2 * n_utt * n_len + 4 * 257 * n * frames
        print(f"ln spectrum x {n:2d}: one launch {one:8.3f} ms ({alg / one / 1e6 / 8000:.3f} of the HBM roofline on {alg / 1e9:.2f} GB)")
        del out
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
