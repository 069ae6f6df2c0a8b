#!/usr/bin/env python3
"""The experimental eight-wave kernel (mfcc_amd/csrc/split_kernel.h, `make split`) against the product's kernel:
MFCC-39 rows of both libraries, bit for bit, on ragged batches (one-frame utterances, lengths that end inside a
tile, several tiles per utterance, more utterances than workgroups), int16 and float32 samples, both frame
geometries and a run-time numcep.  With --time also their launch times on the benchmark batch.

    python tools/split_check.py [--time]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from mfcc_amd import _capi as capi  # noqa: E402
import mfcc_amd.frontend as F  # noqa: E402

PRODUCT = capi.LIB_PATH
SPLIT = os.path.join(ROOT, "build", "variants", "libmfcc_split.so")


def use(path):
    capi._lib = None
    capi.LIB_PATH = path
    F._plan_cache.clear()
    capi.load()


def rows(path, audio, lengths, **cfg):
    use(path)
    fe = F.MFCC("self5", **cfg)
    lay = fe.layout(lengths)
    out = fe.forward(audio, lay)
    torch.cuda.synchronize()
    return out.clone()


def main():
    if not os.path.exists(SPLIT):
        print("split library missing: make split")
        return 2
    rng = np.random.default_rng(42)
    cases = []
    lengths = [1, 399, 400, 401, 560, 4000, 48000, 160 * 303 + 400, 16000 * 11 + 7, 1000] + [int(v) for v in rng.integers(300, 90000, size=700)]
    cases.append(("ragged int16, 16 kHz", lengths, np.int16, {}))
    cases.append(("ragged float32, 16 kHz", lengths[:200], np.float32, {}))
    cases.append(("ragged int16, 8 kHz", [max(1, n // 2) for n in lengths[:300]], np.int16, {"fs": 8000, "num_fft": 512}))
    cases.append(("numcep 12", lengths[:150], np.int16, {"numcep": 12}))
    cases.append(("magnitude spectrum, mel floor", lengths[:150], np.int16, {"spectrum": "mag", "floor_mel": True, "preemph": None}))
    cases.append(("one long utterance", [16000 * 120], np.int16, {}))
    cases.append(("three utterances", [5000, 6000, 7000], np.int16, {}))
    for name, ls, dt, cfg in cases:
        audio = np.concatenate([np.clip(rng.standard_normal(n) * 3000, -32768, 32767) for n in ls])
        t = torch.from_numpy(audio.astype(dt)).cuda()
        a = rows(PRODUCT, t, ls, **cfg)
        b = rows(SPLIT, t, ls, **cfg)
        same = torch.equal(a, b) or bool(((a == b) | (torch.isnan(a) & torch.isnan(b))).all())
        print(f"{name:32s} {a.shape[0]:8d} rows  {'bit-identical' if same else 'DIFFERENT'}", flush=True)
        if not same:
            bad = (~((a == b) | (torch.isnan(a) & torch.isnan(b)))).nonzero()
            print("   first differences at (row, column):", bad[:5].tolist())
            return 1
    if "--time" in sys.argv:
        utts = 10000
        audio = (torch.randn(utts, 48000, device="cuda") * 3000).clamp(-32768, 32767).to(torch.int16).view(-1)
        ctx = []
        for path in (PRODUCT, SPLIT):
            use(path)
            fe = F.MFCC("self5")
            lay = fe.layout([48000] * utts)
            out = torch.empty((lay.total_frames, 39), dtype=torch.float32, device="cuda")
            for _ in range(30):
                fe.forward(audio, lay, out=out)
            torch.cuda.synchronize()
            ctx.append((path, fe, lay, out, []))
        for _ in range(5):
            for path, fe, lay, out, ms in ctx:
                use(path)
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(50):
                    fe.forward(audio, lay, out=out)
                e.record()
                torch.cuda.synchronize()
                ms.append(s.elapsed_time(e) / 50)
        for path, fe, lay, out, ms in ctx:
            print(f"{os.path.basename(path):24s} min {min(ms):.4f} ms  median {sorted(ms)[len(ms) // 2]:.4f} ms per launch of 10 000 x 3 s")
    print("OK")
    return 0


if __name__ == "__main__":
    sys.exit(main())
