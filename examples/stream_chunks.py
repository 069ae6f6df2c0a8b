#!/usr/bin/env python3
"""Live-style use: audio arrives in 10 ms hops (the microphone queue of get_mic_input_13.py:28-65
delivers blocks like these); rows come out as soon as their delta context is complete.

    python examples/stream_chunks.py speech.wav [--hop 160] [--streams 4]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mfcc_amd  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("audio")
    ap.add_argument("--hop", type=int, default=160)
    ap.add_argument("--streams", type=int, default=1)
    args = ap.parse_args()
    import scipy.io.wavfile as wav
    rate, audio = wav.read(args.audio)
    audio = torch.from_numpy(np.tile(audio[None, :], (args.streams, 1))).cuda()
    s = mfcc_amd.StreamingMFCC("self5", n_streams=args.streams, max_chunk=args.hop)
    rows, t0 = [], time.perf_counter()
    for pos in range(0, audio.shape[1], args.hop):
        rows.append(s.push(audio[:, pos:pos + args.hop]).clone())
    rows.append(s.flush().clone())
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    feats = torch.cat(rows, dim=1)
    offline = mfcc_amd.MFCC("self5")(audio).reshape(args.streams, -1, 39)
    print(f"{feats.shape[1]} rows per stream in {dt * 1e3:.1f} ms "
          f"({audio.shape[1] / rate / dt:.0f} x real time per stream); "
          f"max |stream - offline| = {(feats - offline).abs().max().item():.2e}")


if __name__ == "__main__":
    main()
