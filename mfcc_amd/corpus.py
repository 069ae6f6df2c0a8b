"""Corpus-scale extraction with corpus-wide mean / variance normalisation: the rank-local part of
"N hours of audio, utterance-sharded over the GPUs of a node, CMVN constants from all of it".

Every rank owns a contiguous block of utterances (``shard_utterances``), extracts MFCC-39 for it
chunk by chunk into one resident feature array, accumulates float64 (n, sum x, sum x^2), combines
those 1 + 2*D doubles with the other ranks (the only exchange: an all-gather over RCCL and a sum in
rank order), and normalises its rows in place with (x + mm) * vv, the form of
anal_recog_nn_liste.m:180 (generalising the per-utterance (x - mean) / std of
inference_ctc_eri_05.py:291).  No collective touches audio or feature rows.
"""
from __future__ import annotations

import numpy as np
import torch

from . import frontend as F


_side_streams = {}


def _side_stream(dev):
    s = _side_streams.get(dev.index)
    if s is None:
        s = _side_streams[dev.index] = torch.cuda.Stream(device=dev)
    return s


def corpus_mfcc_cmvn(fe, audio, lengths, chunk_utts=10000, out=None, group=None, ddof=0, events=None, static_scratch=False,
                     per_chunk_sums=False, overlap=False, fused_sums=False):
    """fe: MFCC extractor; audio: this rank's packed device samples; lengths: its utterance
    lengths.  Returns (feats [sum F, D] normalised in place, sums float64[1 + 2 D] of the whole
    corpus, mm, vv).  ``events``: optional dict that receives per-phase lists of (start, end)
    torch.cuda.Event pairs recorded on the current stream (phases: forward, accumulate, exchange,
    apply).  ``per_chunk_sums``: accumulate the statistics after every chunk (while its rows are still in
    the caches) instead of in one pass after the last chunk.  ``overlap``: the same per-chunk accumulation on a
    second stream, ordered behind its chunk's extraction by an event, so that the memory-bound statistics pass
    runs beside the arithmetic-bound extraction of the next chunk; the sums are added chunk by chunk in chunk
    order either way (equal inputs give equal bits).  ``fused_sums``: the extraction kernel itself adds up the rows it
    writes (MFCC.forward_sums: float32 per tile, float64 across tiles, fixed order), so no pass reads the rows back for
    the statistics; plans of the specialised kernel."""
    lengths = np.asarray(lengths, dtype=np.int64).reshape(-1)
    plan = fe.plan
    counts = np.array([plan.frames_for_length(int(n)) for n in lengths], dtype=np.int64) if len(set(lengths.tolist())) > 1 \
        else np.full(len(lengths), plan.frames_for_length(int(lengths[0])) if len(lengths) else 0, dtype=np.int64)
    total_frames = int(counts.sum())
    d = plan.out_width("mfcc39")
    dev = plan.device
    if out is None:
        out = torch.empty((total_frames, d), dtype=torch.float32, device=dev)
    elif out.shape[0] != total_frames or out.shape[1] < d:
        raise ValueError(f"out must hold {total_frames} rows of at least {d} floats")
    feats = out[:, :d]
    sums = torch.zeros(1 + 2 * d, dtype=torch.float64, device=dev)

    def mark(phase):
        if events is None:
            return None
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        events.setdefault(phase, []).append((a, b))
        a.record()
        return b

    if fused_sums and (overlap or per_chunk_sums):
        raise ValueError("fused_sums replaces the statistics pass: not together with per_chunk_sums or overlap")
    ws = None
    s_pos = f_pos = 0
    with torch.cuda.device(dev):
        main = torch.cuda.current_stream(dev)
        side = _side_stream(dev) if overlap else None
        if side is not None:
            side.wait_stream(main)      # `sums` was zeroed on the main stream
        for u0 in range(0, len(lengths), chunk_utts):
            ls = lengths[u0:u0 + chunk_utts]
            ns, nf = int(ls.sum()), int(counts[u0:u0 + chunk_utts].sum())
            lay = fe.layout(ls)
            view = out[f_pos:f_pos + nf]
            e = mark("forward")
            if fused_sums:
                need = fe.forward_sums_workspace(lay)
                if ws is None or ws.numel() < need:
                    ws = torch.empty(need, dtype=torch.float32, device=dev)
                fe.forward_sums(audio[s_pos:s_pos + ns], lay, sums, out=view, workspace=ws)
            else:
                fe.forward(audio[s_pos:s_pos + ns], lay, out=view)
            if e: e.record()
            if side is not None:
                side.wait_stream(main)  # behind this chunk's extraction; the next chunk's runs beside it
                with torch.cuda.stream(side):
                    e = mark("accumulate")
                    F.cmvn_accumulate(view[:, :d], sums, static=static_scratch)
                    if e: e.record()
            elif per_chunk_sums:
                e = mark("accumulate")
                F.cmvn_accumulate(view[:, :d], sums, static=static_scratch)
                if e: e.record()
            s_pos += ns
            f_pos += nf
        if side is not None:
            main.wait_stream(side)
        elif not per_chunk_sums and not fused_sums:
            # the rows are all resident: one pass over them (the extraction launches then run back to back)
            e = mark("accumulate")
            F.cmvn_accumulate(feats, sums, static=static_scratch)
            if e: e.record()
        e = mark("exchange")
        F.cmvn_allreduce(sums, group=group)
        mm, vv = F.cmvn_constants(sums, ddof=ddof)
        if e: e.record()
        e = mark("apply")
        F.cmvn_apply(feats, mm, vv)
        if e: e.record()
    return feats, sums, mm, vv
