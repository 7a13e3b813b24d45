"""Ground-truth loading mirror of /root/reference/BSD_metrics/groundtruth.py:16-50 (host side,
scipy.io) plus a packed-fixture reader (the .npz form the tests use, which
needs neither .mat parsing nor the per-id directory scan of groundtruth.py:44-48)."""
from __future__ import annotations

import os

import numpy as np


def get_segmentation(path, filename):
    """groundtruth.py:16-29: the `Segmentation` field of every annotator in a BSD500 .mat file."""
    from scipy.io import loadmat
    f = loadmat(path + filename)
    return [img[0][0][0] for img in f['groundTruth'][0]]


def get_segment_from_filename(filename, path="./data/truth/"):
    """groundtruth.py:33-50: look the id up in every split directory under ``path``."""
    filename = filename + '.mat'
    segments = []
    for folder in os.listdir(path):
        if filename in os.listdir(path + folder + "/"):
            segments.extend(get_segmentation(path + folder + "/", filename))
    return segments


def load_packed(npz_path):
    """{id: (image uint8 (H,W,3), [annotator maps uint16 (H,W)])} from a packed fixture
    (tests/golden/bsd_inputs.npz, written by tests/golden/make_inputs.py)."""
    z = np.load(npz_path)
    out = {}
    for i in z["ids"]:
        i = str(i)
        out[i] = (z["img_" + i], [z["seg_%s_%d" % (i, a)] for a in range(int(z["nseg_" + i]))])
    return out
