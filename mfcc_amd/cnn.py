"""The CNN scripts' flavour of the .pat plumbing (CNN_DenseNet40_6.py:59-171 and its copies): the
same function names as the MLP scripts, but ``slidingWin`` keeps each window as the [C, N] slice
``paddeddata[:, pos:pos+framenum]`` (:98) and *applies* the shuffled window order (:91-97), and
``createMinibatchData`` builds ``[minibatch, C, N, 1]`` (:124-131).  The windows are cut on the
device in one launch (tile layout of the stacking kernel); the shuffle is its row-index argument.
"""
from __future__ import annotations

import numpy as np

from . import datasets as D
from . import frontend as F
from .datasets import (createMinibatchLabel, createRandomvec, loadNamefiles, randomShuffleData)  # noqa: F401  (same in both flavours)
from .frontend import loadPatData, padUtterance  # noqa: F401  (CNN_DenseNet40_6.py:59-84, identical to the MLP ones)


def slidingWin(framenum, paddedlabel, paddeddata, randomint=-11):
    """CNN_DenseNet40_6.py:86-103 -> (labellist, datalist); datalist is a [F, C, N] array that
    iterates tile by tile.  ``randomint != -11`` visits the windows in the order of
    createRandomvec(randomint, F)."""
    paddedlabel = np.asarray(paddedlabel)
    x, was_numpy = F._as_rows(paddeddata)
    half = int((framenum - 1) / 2)
    nwin = paddedlabel.shape[0] - framenum + 1
    order = np.arange(nwin) if randomint == -11 else createRandomvec(randomint, nwin)
    labellist = [paddedlabel[pos + half] for pos in order]
    padded = F._to_dev(x, F.torch.float32).t().contiguous()            # [Fp, C]
    # window pos of the padded data is centred on padded row pos + half; no clamping happens there
    data = F.stack_context(padded, framenum, layout="tile", order=np.asarray(order, dtype=np.int64) + half)
    return labellist, F._ret(data, was_numpy)


def createMinibatchData(minibatchsize, framenum, databuffer):
    """CNN_DenseNet40_6.py:124-131: the first ``minibatchsize`` tiles as float64 [mb, C, N, 1]."""
    buf = np.asarray(databuffer[:minibatchsize])
    minibatchdata = np.zeros(shape=(minibatchsize, buf.shape[1], framenum))
    minibatchdata[:] = buf
    return minibatchdata[:, :, :, np.newaxis]


def Preprocess(framenum, randbatnum, dataset_name, path, randomint):
    """CNN_DenseNet40_6.py:143-160."""
    datafile = loadPatData(path + D.data_dir + dataset_name[randbatnum], 12)
    labelfile = np.loadtxt(path + D.label_dir + dataset_name[randbatnum][:-4] + ".txt")
    paddedlabel, paddeddata = padUtterance(framenum, labelfile, datafile)
    return slidingWin(framenum, paddedlabel, paddeddata, randomint)


def Databuffer(buffersamples, framenum, dataset_name, path, randomint):
    """CNN_DenseNet40_6.py:162-171 -> (label_buffer, data_buffer [sum F, C, N]) for the first
    ``buffersamples`` utterances: one host-to-device copy, one launch, one copy back.  With
    ``randomint != -11`` every utterance's windows come in the order of
    createRandomvec(randomint, F_u), as slidingWin gives them."""
    packed, counts, labels = D.load_utterances(buffersamples, dataset_name, path)
    if not counts:
        return [], np.zeros((0, 0, framenum), np.float32)
    order, label_buffer, pos = [], [], 0
    for n, lab in zip(counts, labels):
        perm = np.arange(n) if randomint == -11 else createRandomvec(randomint, n)
        order.append(pos + perm)
        label_buffer += [lab[i] for i in perm]
        pos += n
    data = F.stack_context(packed, framenum, frame_counts=counts, layout="tile", order=np.concatenate(order))
    return label_buffer, data
