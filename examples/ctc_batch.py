#!/usr/bin/env python3
"""Feature side of create_minibatch_ctc (inference_ctc_eri_05.py:283-300): a list of .wav files ->
normalised, zero-padded [B, T, 39] features + sequence lengths, computed on the GPU.

    python examples/ctc_batch.py a.wav b.wav ... [--time-steps 700]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mfcc_amd  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("wavs", nargs="+")
    ap.add_argument("--time-steps", type=int, default=None)
    args = ap.parse_args()
    import scipy.io.wavfile as wav
    audios = [wav.read(p)[1] for p in args.wavs]
    x, seq_len = mfcc_amd.ctc_feature_batch(audios, time_steps=args.time_steps)
    print("features", tuple(x.shape), "seq_length", seq_len.tolist())


if __name__ == "__main__":
    main()
