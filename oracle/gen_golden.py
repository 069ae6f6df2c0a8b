#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/ from the REFERENCE's own code.

TEST INFRASTRUCTURE ONLY.  Run in the build container (needs /root/reference;
never runs on the GPU box):

    MPLBACKEND=Agg python oracle/gen_golden.py

What it executes from the reference:
  * mfcc-self5.py / mfcc-self15.py are loaded with importlib (the hyphenated
    names rule out ``import``); a stub ``sounddevice`` module is pre-inserted
    because both files import it at top level (self5:16) and it is not
    installed.  ``main`` is never called (guarded, self5:259); the pipeline
    functions are driven in main's order (self5:211-229, self15:228-246).
  * From MLP_5hidlayer_40.py only the function definitions loadPatData,
    padUtterance, slidingWin, createRandomvec (lines 61-112) are compiled
    (the module trains a TensorFlow model at import time, so it cannot be
    imported whole).

Only numeric inputs/outputs are written; no reference source is copied.
"""
import ast
import importlib.util
import io
import os
import sys
import types
import zipfile

import numpy as np
import scipy.io.wavfile as wavfile

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def load_reference(fname, modname):
    sys.modules.setdefault("sounddevice", types.ModuleType("sounddevice"))
    spec = importlib.util.spec_from_file_location(modname, os.path.join(REF, fname))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def run_pipeline(mod, audio, variant, fs=16000, frame_time=None, step_time=None):
    """Drive the reference functions exactly as its main() does."""
    frame_time = mod.frame_time if frame_time is None else frame_time
    step_time = mod.step_time if step_time is None else step_time
    audio_pre = mod.preemphasis(audio)
    src = audio_pre if variant == "self5" else audio  # self15:230 feeds the raw signal
    magfft, powfft, win, nfft = mod.winfft(src, fs, frame_time, step_time)
    bank = mod.melbanks(fs, mod.num_bins, mod.low_freq, nfft)
    spec = powfft if variant == "self5" else magfft  # self15:234 feeds magnitudes
    with np.errstate(divide="ignore", invalid="ignore"):
        mel_l, en_l = mod.melspec(spec, bank)
        cep_l = mod.mfcc(mel_l, numcep=13)
        cep = np.array(cep_l)
        lift_l = mod.dctopt(cep_l, mod.lift_factor)
        lifted = np.array(lift_l)  # copy BEFORE staticfeatures mutates the list (self5:164)
        static = mod.staticfeatures(en_l, lift_l)
        d = mod.delta(static)
        dd = mod.delta(d)
    out = mod.spectrograms(static, d, dd)
    return dict(signal=np.asarray(audio_pre, dtype=np.float64), mag=np.array(magfft),
                pow=np.array(powfft), win=win, nfft=np.int64(nfft), melbank=bank,
                logmel=np.array(mel_l), loge=np.array(en_l, dtype=np.float64), cep=cep,
                lifted=lifted, static=static, delta=d, delta2=dd, out=out)


def sample_rows(n):
    rows = sorted(set(list(range(min(8, n))) + list(range(0, n, 16)) + list(range(max(0, n - 8), n))))
    return np.array(rows, dtype=np.int64)


def synth(seed, length):
    rng = np.random.default_rng(seed)
    return np.clip(rng.standard_normal(length) * 3000, -32768, 32767).astype(np.int16)


def main():
    os.makedirs(OUT, exist_ok=True)
    ref5 = load_reference("mfcc-self5.py", "ref_self5")
    ref15 = load_reference("mfcc-self15.py", "ref_self15")
    mods = {"self5": ref5, "self15": ref15}

    # ---- 1. the one real utterance in the reference (CTC/LDC93S1.wav) -------
    fs, audio = wavfile.read(os.path.join(REF, "CTC", "LDC93S1.wav"))
    assert fs == 16000 and audio.dtype == np.int16 and audio.shape == (46797,)
    pack = dict(audio=audio, fs=np.int64(fs))
    for variant, mod in mods.items():
        st = run_pipeline(mod, audio, variant)
        rows = sample_rows(st["out"].shape[0])
        pack[f"{variant}_rows"] = rows
        for k in ("mag", "pow", "logmel", "cep", "lifted"):
            pack[f"{variant}_{k}_rows"] = st[k][rows]
        pack[f"{variant}_logmel"] = st["logmel"]
        for k in ("loge", "static", "delta", "delta2", "out"):
            pack[f"{variant}_{k}"] = st[k]
        pack[f"{variant}_signal_head"] = st["signal"][:512]
    pack["melbank"] = ref5.melbanks(16000, 26, 0, 512)
    pack["win"] = st["win"]
    np.savez_compressed(os.path.join(OUT, "ldc93s1.npz"), **pack)

    # ---- 2. seeded synthetic utterances, frame-count edge cases ----------------
    pack = {}
    lengths = [1, 2, 399, 400, 401, 560, 561, 4000, 48000]
    pack["lengths"] = np.array(lengths, dtype=np.int64)
    pack["seeds"] = np.array([100 + i for i in range(len(lengths))], dtype=np.int64)
    for i, length in enumerate(lengths):
        audio = synth(100 + i, length)
        pack[f"audio_{length}"] = audio
        for variant, mod in mods.items():
            st = run_pipeline(mod, audio, variant)
            pack[f"{variant}_out_{length}"] = st["out"]
            if length <= 4000:
                pack[f"{variant}_logmel_{length}"] = st["logmel"]
                pack[f"{variant}_loge_{length}"] = st["loge"]
            if length in (401, 561):
                pack[f"{variant}_pow_{length}"] = st["pow"]
                pack[f"{variant}_mag_{length}"] = st["mag"]
    np.savez_compressed(os.path.join(OUT, "synthetic.npz"), **pack)

    # ---- 3. digital silence, near-silence and full-scale input --------------------
    pack = {}
    cases = {
        "zeros": np.zeros(1000, dtype=np.int16),
        "tailzeros": np.concatenate([synth(7, 700), np.zeros(900, dtype=np.int16)]),
        "fullscale": np.tile(np.array([32767, -32768], dtype=np.int16), 600),
        "dc": np.full(1200, 1000, dtype=np.int16),
    }
    for name, audio in cases.items():
        pack[f"audio_{name}"] = audio
        for variant, mod in mods.items():
            st = run_pipeline(mod, audio, variant)
            pack[f"{variant}_out_{name}"] = st["out"]
            pack[f"{variant}_logmel_{name}"] = st["logmel"]
            pack[f"{variant}_loge_{name}"] = st["loge"]
    np.savez_compressed(os.path.join(OUT, "special.npz"), **pack)

    # ---- 4. float32 input (audio / 32768) ---------------------------------------
    pack = {}
    audio = (synth(42, 3000).astype(np.float32) / np.float32(32768.0)).astype(np.float32)
    pack["audio"] = audio
    for variant, mod in mods.items():
        st = run_pipeline(mod, audio, variant)
        pack[f"{variant}_out"] = st["out"]
        pack[f"{variant}_logmel"] = st["logmel"]
    np.savez_compressed(os.path.join(OUT, "float_input.npz"), **pack)

    # ---- 5. per-stage functions on arbitrary inputs --------------------------------
    rng = np.random.default_rng(5)
    pack = {}
    feats = rng.standard_normal((37, 13)) * 10
    pack["delta_in"] = feats
    pack["delta_out"] = ref5.delta(feats)
    pack["delta_out_pad3"] = ref5.delta(feats, 3)
    one = rng.standard_normal((1, 13))
    pack["delta_in_1"] = one
    pack["delta_out_1"] = ref5.delta(one)
    lm = rng.standard_normal((9, 26)) * 4 + 10
    pack["dct_in"] = lm
    pack["dct_out12"] = np.array(ref5.mfcc(list(lm)))
    pack["dct_out13"] = np.array(ref5.mfcc(list(lm), numcep=13))
    pack["lift_out"] = np.array(ref5.dctopt(list(pack["dct_out13"]), 22))
    pack["lift_out_0"] = np.array(ref5.dctopt(list(pack["dct_out13"]), 0))
    pack["melbank_40"] = ref5.melbanks(16000, 40, 0, 512)
    pack["melbank_26_low300"] = ref5.melbanks(16000, 26, 300, 512)
    pack["melbank_8k_26_256"] = ref5.melbanks(8000, 26, 0, 256)
    sig = synth(9, 50)
    pack["preemph_in"] = sig
    pack["preemph_out"] = ref5.preemphasis(sig)
    pack["preemph_out_095"] = ref5.preemphasis(sig, 0.95)
    np.savez_compressed(os.path.join(OUT, "stages.npz"), **pack)

    # ---- 6. .pat container + context stacking (MLP_5hidlayer_40.py:61-112) ----------
    with open(os.path.join(REF, "MLP_5hidlayer_40.py"), "r", encoding="utf-8", errors="replace") as fh:
        tree = ast.parse(fh.read())
    wanted = {"loadPatData", "padUtterance", "slidingWin", "createRandomvec"}
    body = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in wanted]
    ns = {"np": np, "os": os, "coeffs": 39}
    exec(compile(ast.Module(body=body, type_ignores=[]), "MLP_5hidlayer_40.py", "exec"), ns)
    pats = zipfile.ZipFile(os.path.join(REF, "39pat.zip"))
    labs = zipfile.ZipFile(os.path.join(REF, "nn_output30.zip"))
    pack = {}
    tmpdir = "/tmp/_golden_pat"
    os.makedirs(tmpdir, exist_ok=True)
    for name in ("is1c0001_001", "is1c0001_031"):
        raw = pats.read(f"39pat/{name}.pat")
        path = os.path.join(tmpdir, name + ".pat")
        with open(path, "wb") as fh:
            fh.write(raw)
        data = ns["loadPatData"](path, 12)  # [C, F]
        labels = np.loadtxt(io.BytesIO(labs.read(f"nn_output30/{name}.txt")))
        pack[f"{name}_pat_bytes"] = np.frombuffer(raw, dtype=np.uint8)
        pack[f"{name}_data"] = data
        pack[f"{name}_labels"] = labels
        for framenum in (1, 11, 17):
            plab, pdat = ns["padUtterance"](framenum, labels, data)
            lablist, veclist = ns["slidingWin"](framenum, plab, pdat, -11)
            pack[f"{name}_stack{framenum}"] = np.array(veclist, dtype=np.float32)
            pack[f"{name}_stacklab{framenum}"] = np.array(lablist)
    np.savez_compressed(os.path.join(OUT, "stacking.npz"), **pack)

    # ---- 7. the CNN scripts' windows: [C, N] tiles in (optionally) shuffled order ------------
    #         (CNN_DenseNet40_6.py:59-131; the file builds a TensorFlow graph at import time, so
    #         only these function definitions are compiled)
    with open(os.path.join(REF, "CNN_DenseNet40_6.py"), "r", encoding="utf-8", errors="replace") as fh:
        tree = ast.parse(fh.read())
    wanted = {"loadPatData", "padUtterance", "slidingWin", "createRandomvec", "createMinibatchData"}
    body = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in wanted]
    assert {n.name for n in body} == wanted
    cns = {"np": np, "os": os, "coeffs": 39}
    exec(compile(ast.Module(body=body, type_ignores=[]), "CNN_DenseNet40_6.py", "exec"), cns)
    pack = {}
    name = "is1c0001_001"
    path = os.path.join(tmpdir, name + ".pat")
    data = cns["loadPatData"](path, 12)
    labels = np.loadtxt(io.BytesIO(labs.read(f"nn_output30/{name}.txt")))
    pack["name"] = np.array(name)
    pack["data"] = data
    pack["labels"] = labels
    for framenum in (11, 17):
        for randomint in (-11, 3):
            plab, pdat = cns["padUtterance"](framenum, labels, data)
            lablist, tiles = cns["slidingWin"](framenum, plab, pdat, randomint)
            tag = f"n{framenum}_r{randomint if randomint >= 0 else 'm11'}"
            pack[f"tiles_{tag}"] = np.array(tiles, dtype=np.float32)          # [F, C, N]
            pack[f"labels_{tag}"] = np.array(lablist)
            pack[f"minibatch_{tag}"] = cns["createMinibatchData"](16, framenum, tiles)   # float64 [16, C, N, 1]
    np.savez_compressed(os.path.join(OUT, "stacking_cnn.npz"), **pack)

    # ---- 8. other frame geometries of winfft (self5:75-84): 8 kHz -> 200/80 samples, 256-point FFT;
    #         44.1 kHz -> 1102/441, 2048-point; 20 ms frames at 16 kHz -> 320/160, 512-point ------
    pack = {}
    geoms = {
        "8k": dict(fs=8000, frame_time=25, step_time=10, lengths=[1, 199, 200, 201, 281, 8000]),
        "44k": dict(fs=44100, frame_time=25, step_time=10, lengths=[1, 1102, 1103, 1544, 44100]),
        "16k_20ms": dict(fs=16000, frame_time=20, step_time=10, lengths=[319, 320, 321, 481, 5000]),
        "16k_32ms_8ms": dict(fs=16000, frame_time=32, step_time=8, lengths=[512, 513, 4000]),
    }
    for gname, gcfg in geoms.items():
        pack[f"{gname}_lengths"] = np.array(gcfg["lengths"], dtype=np.int64)
        pack[f"{gname}_cfg"] = np.array([gcfg["fs"], gcfg["frame_time"], gcfg["step_time"]], dtype=np.float64)
        for i, length in enumerate(gcfg["lengths"]):
            audio = synth(300 + i, length)
            pack[f"{gname}_audio_{length}"] = audio
            for variant, mod in mods.items():
                st = run_pipeline(mod, audio, variant, gcfg["fs"], gcfg["frame_time"], gcfg["step_time"])
                pack[f"{gname}_{variant}_out_{length}"] = st["out"]
                pack[f"{gname}_{variant}_logmel_{length}"] = st["logmel"]
                pack[f"{gname}_{variant}_loge_{length}"] = st["loge"]
                pack[f"{gname}_nfft"] = st["nfft"]
                if i == 1:
                    pack[f"{gname}_{variant}_spec_{length}"] = st["pow"] if variant == "self5" else st["mag"]
    np.savez_compressed(os.path.join(OUT, "geometry.npz"), **pack)

    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
