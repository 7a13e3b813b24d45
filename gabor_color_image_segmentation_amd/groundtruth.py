"""Ground-truth loading mirror of /root/reference/BSD_metrics/groundtruth.py:16-50 (host side,
scipy.io) plus packed-fixture readers (.npz forms that need neither .mat parsing
nor the per-id scan of every split directory, groundtruth.py:44-48): ``load_packed`` for the small image +
truth fixtures, ``PackedTruth`` for the whole-dataset pack written by tools/pack_bsd_truth.py (all 500 ids,
2 696 annotator maps, 7 MB) including the ragged device layout the batched GPU scorer takes."""
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


class PackedTruth:
    """All annotator maps of a set of BSD ids from ONE file (tools/pack_bsd_truth.py ran the reference's own loader,
    groundtruth.py:33-50, over the dataset). ``pt[id]`` -> list of (H,W) uint16 maps, as get_segment_from_filename."""

    def __init__(self, npz_path):
        z = np.load(npz_path)
        self.ids = [str(i) for i in z["ids"]]
        self.hw = z["hw"].astype(int)
        self.first = z["first"].astype(np.int64)
        self.offs = z["offs"].astype(np.int64)
        self.data = z["data"]
        self._index = {i: n for n, i in enumerate(self.ids)}

    def __len__(self):
        return len(self.ids)

    def __contains__(self, i):
        return str(i) in self._index

    def shape(self, i):
        return tuple(self.hw[self._index[str(i)]])

    def n_annotators(self, i):
        n = self._index[str(i)]
        return int(self.first[n + 1] - self.first[n])

    def __getitem__(self, i):
        n = self._index[str(i)]
        h, w = self.hw[n]
        return [self.data[self.offs[t]:self.offs[t + 1]].reshape(h, w).astype(np.uint16)
                for t in range(self.first[n], self.first[n + 1])]

    def stack(self, ids):
        """Ragged batch layout of gcs_boundary_counts_batch / gcs_region_counts_batch for images of ONE shape:
        (truth uint16 [T,H,W], first int32 [B+1], img_of int32 [T], n_truth list of max+1 per map)."""
        ids = [str(i) for i in ids]
        shapes = {self.shape(i) for i in ids}
        if len(shapes) != 1:
            raise ValueError(f"one batch needs one image shape, got {sorted(shapes)}")
        maps, first, img_of = [], [0], []
        for b, i in enumerate(ids):
            m = self[i]
            maps.extend(m)
            img_of.extend([b] * len(m))
            first.append(first[-1] + len(m))
        truth = np.stack(maps)
        return truth, np.array(first, np.int32), np.array(img_of, np.int32), [int(m.max()) + 1 for m in maps]

    def to_device(self, ids, device="cuda"):
        """The annotator maps of ``ids`` (one image shape) resident on ``device`` in the scorer's form: uint8 maps, bit planes
        of their thick boundaries and of the 5x5 dilation, the boundary counts (``evaluate_gpu.DeviceTruth``). What
        /root/reference/BSD_metrics/metrics.py:48-49 and groundtruth.py:44-48 redo per image happens here once per id."""
        from .evaluate_gpu import DeviceTruth
        return DeviceTruth(*self.stack(ids), device=device)
