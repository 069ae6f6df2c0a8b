#!/usr/bin/env python3
"""The body of main() of mfcc-self5.py (:188-229) / mfcc-self15.py (:205-246) on the MI355X
functions: same names, same call order, one .wav (or headerless int16 .raw) file in, the
[F, 39] spectrogram out (.npy and, optionally, a .pat file for the training scripts).

    python examples/mfcc_self5_main.py speech.wav [--variant self15] [--pat out.pat] [--fused]
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mfcc_amd import *  # noqa: F401,F403  (the reference's function names and module globals)
import mfcc_amd  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("audio")
    ap.add_argument("--variant", default="self5", choices=("self5", "self15"))
    ap.add_argument("--pat", default=None)
    ap.add_argument("--fused", action="store_true", help="one launch instead of the stage-by-stage chain")
    args = ap.parse_args()

    if args.audio.endswith(".wav"):
        import scipy.io.wavfile as wav
        rate, audio = wav.read(args.audio)          # mfcc-self5.py:196
        assert rate == fs, f"sample rate {rate}, the pipeline is set up for {fs}"
    else:
        audio = loadrawdata(args.audio)             # mfcc-self5.py:32-36

    if args.fused:
        spec = mfcc_amd.MFCC(args.variant)(audio)
    else:
        audio_pre = preemphasis(audio)                                               # :211
        sig = audio_pre if args.variant == "self5" else audio                       # self15 skips it, :230
        magfft_list, powfft_list, win, num_fft = winfft(sig, fs)                     # :213
        melbank = melbanks(fs, num_bins, low_freq, num_fft)                          # :215
        melspec_list, energy_list = melspec(powfft_list if args.variant == "self5" else magfft_list, melbank)  # :217
        mfcc_list = mfcc(melspec_list, 13)                                           # :219
        mfcclift_list = dctopt(mfcc_list, lift_factor)                               # :221
        features = staticfeatures(energy_list, mfcclift_list)                        # :223
        delta_features = delta(features, 2)                                          # :225
        delta_delta_features = delta(delta_features, 2)                              # :227
        spec = spectrograms(features, delta_features, delta_delta_features)          # :229
    spec = np.asarray(spec)
    out = os.path.splitext(args.audio)[0] + f".{args.variant}.mfcc39.npy"
    np.save(out, spec)
    print(f"{args.audio}: {spec.shape[0]} frames x {spec.shape[1]} coefficients -> {out}")
    if args.pat:
        writePatData(args.pat, spec)
        print(f"wrote {args.pat}")


if __name__ == "__main__":
    main()
