"""Host-side counterparts of the data plumbing around the features in MLP_5hidlayer_40.py
(``Preprocess`` / ``Databuffer`` and friends, :56-172) so that the training scripts can consume
features produced here through their own on-disk format (.pat files + label .txt files) with the
context stacking done on the device.  Same names, argument order and return structure; "lists of
vectors" are 2-D arrays that iterate vector by vector.
"""
from __future__ import annotations

import os

import numpy as np

from . import frontend as F

# module-level globals the reference's functions read (MLP_5hidlayer_40.py:222-247)
data_dir = "00_data/pattern_dft_16k_norm/"
label_dir = "00_data/nn_output_tri/"


def loadNamefiles(filenames):
    """MLP_5hidlayer_40.py:56-59."""
    return np.genfromtxt(filenames, delimiter=" ", dtype="str")


def createRandomvec(randomint, epochlength):
    """MLP_5hidlayer_40.py:107-112: the permutation of arange(epochlength) under seed randomint."""
    np.random.seed(randomint)
    randomvec = np.arange(epochlength)
    np.random.shuffle(randomvec)
    return randomvec


def randomShuffleData(randomint, dataset_name):
    """MLP_5hidlayer_40.py:114-124."""
    randomvec = createRandomvec(randomint, len(dataset_name))
    return [dataset_name[randomvec[pos]] for pos in range(len(dataset_name))]


def Preprocess(framenum, randbatnum, dataset_name, path, randomint):
    """MLP_5hidlayer_40.py:144-161: load one .pat file and its label file, pad, window.
    Returns (labellist, datavectors [F, framenum*C])."""
    datafilename = path + data_dir + dataset_name[randbatnum]
    datafile = F.loadPatData(datafilename, 12)
    labelfile = np.loadtxt(path + label_dir + dataset_name[randbatnum][:-4] + ".txt")
    paddedlabel, paddeddata = F.padUtterance(framenum, labelfile, datafile)
    return F.slidingWin(framenum, paddedlabel, paddeddata, randomint)


def load_utterances(buffersamples, dataset_name, path):
    """Read the first ``buffersamples`` .pat files and label files (MLP_5hidlayer_40.py:150-156)
    into one packed host array: (feats [sum F, C] float32, frame_counts, labels per utterance).
    A label file one entry longer than its utterance loses the last entry, as padUtterance
    does (:72-73)."""
    feats, counts, labels = [], [], []
    for pos in range(buffersamples):
        data = F.loadPatData(path + data_dir + dataset_name[pos], 12)          # [C, F]
        lab = np.loadtxt(path + label_dir + dataset_name[pos][:-4] + ".txt")
        if lab.shape[0] != data.shape[1]:
            lab = lab[0:-1]
        feats.append(np.ascontiguousarray(data.T))
        counts.append(data.shape[1])
        labels.append(lab)
    packed = np.concatenate(feats) if feats else np.zeros((0, F.coeffs), np.float32)
    return packed, counts, labels


def Databuffer(buffersamples, framenum, dataset_name, path, randomint):
    """MLP_5hidlayer_40.py:163-172: the first ``buffersamples`` utterances, windowed and
    concatenated -> (labelbuffer, datavectorbuffer [sum F, framenum*C]).  The whole buffer is one
    host-to-device copy, one stacking launch over the ragged batch and one copy back; the values
    equal calling Preprocess per utterance.  ``randomint`` is accepted and, like the permutation
    the reference computes from it (:94-96), not applied."""
    packed, counts, labels = load_utterances(buffersamples, dataset_name, path)
    if not counts:
        return [], np.zeros((0, 0), np.float32)
    labelbuffer = [v for lab in labels for v in lab]   # centre label of window t = label of frame t
    return labelbuffer, F.stack_context(packed, framenum, frame_counts=counts)


def createMinibatchData(minibatchsize, framenum, datavectorbuffer):
    """MLP_5hidlayer_40.py:126-132: the first ``minibatchsize`` vectors as float64 [mb, N*C]."""
    buf = np.asarray(datavectorbuffer)
    minibatchdata = np.zeros(shape=(minibatchsize, buf.shape[1]))
    minibatchdata[:] = buf[:minibatchsize]
    return minibatchdata


def createMinibatchLabel(minibatchsize, classnum, labelbuffer):
    """MLP_5hidlayer_40.py:134-142: one-hot rows, class k at column k-1."""
    minibatchlabel = np.zeros(shape=(minibatchsize, classnum))
    for labelpos in range(minibatchsize):
        minibatchlabel[labelpos][int(labelbuffer[labelpos]) - 1] = 1.0
    return minibatchlabel


def energy_last(feats, numcep=13):
    """[F, 3*numcep] with the log-energy FIRST in each block (the numpy routine,
    mfcc-self5.py:164) -> log-energy LAST in each block, the HTK MFCC_E_D_A order of the
    reference's shipped 39-coefficient .pat files."""
    feats = np.asarray(feats)
    idx = np.concatenate([np.r_[b * numcep + 1:(b + 1) * numcep, b * numcep] for b in range(feats.shape[1] // numcep)])
    return feats[:, idx]


def write_pat_corpus(path, names, feats, frame_counts, energy_position="first", numcep=13):
    """Write one .pat file per utterance into ``path + data_dir`` from a packed feature array
    [sum F, C] (device tensor or numpy): one device-to-host copy for the whole batch, then the
    12-byte-header files the training scripts read (MLP_5hidlayer_40.py:61-68, :152)."""
    if energy_position not in ("first", "last"):
        raise ValueError("energy_position must be 'first' or 'last'")
    host = feats.detach().cpu().numpy() if hasattr(feats, "detach") else np.asarray(feats)
    if int(np.sum(frame_counts)) != host.shape[0] or len(names) != len(frame_counts):
        raise ValueError("names / frame_counts do not match the feature rows")
    os.makedirs(path + data_dir, exist_ok=True)
    pos = 0
    for name, n in zip(names, frame_counts):
        block = host[pos:pos + n]
        F.writePatData(path + data_dir + name, energy_last(block, numcep) if energy_position == "last" else block)
        pos += n
