"""Scoring mirror of the reference's ``metrics`` class, on scipy.ndimage (host side).

Same constructor, method names, result keys and error behaviour as
/root/reference/BSD_metrics/metrics.py:18-255 so that ``script.py:36-38`` reads the same:

    m = metrics(img, labels, segments); m.set_metrics(); m.display_metrics()

Differences, all additive: ``fmeasure`` (the reference computes no F)
and vectorised implementations of the two per-pixel Python loops (metrics.py:120-126,
168-180). scikit-image is not available beside torch, so ``find_boundaries`` /
``dilation(rectangle(5,5))`` are restated on scipy.ndimage (verified bit-identical, and
pinned by tests/golden/scoring_golden.json which the reference class itself produced).
This module is product code: it never imports ``oracle``.
"""
from __future__ import annotations

from math import pi

import numpy as np
from scipy import ndimage as ndi

_CROSS = ndi.generate_binary_structure(2, 1)


def find_boundaries(lab: np.ndarray) -> np.ndarray:
    """skimage.segmentation.find_boundaries(lab) defaults (mode='thick', connectivity=1),
    as called at metrics.py:49,69,88,157: 3x3-cross max != min, reflect border."""
    lab = np.asarray(lab)
    return ndi.grey_dilation(lab, footprint=_CROSS) != ndi.grey_erosion(lab, footprint=_CROSS)


def _dilate(mask: np.ndarray, size: int) -> np.ndarray:
    """skimage.morphology.dilation(mask, rectangle(size, size)) for boolean masks."""
    return ndi.grey_dilation(mask.astype(np.uint8), footprint=np.ones((size, size), bool)).astype(bool)


class metrics:
    """Compute the metrics of a segmentation against the BSD manual segmentations."""

    def __init__(self, img, lb, segments_truth):
        # metrics.py:25-51
        self.img = img
        self.lb = np.asarray(lb).astype('int')
        self.nx, self.ny = self.lb.shape
        self.segments_truth = segments_truth
        self.img_truth = [find_boundaries(s) for s in self.segments_truth]
        self.n_segments = np.max(self.lb) + 1

    # ---- boundary recall / precision (metrics.py:58-96)
    def set_boundary_recall(self, size=5):
        bd = _dilate(find_boundaries(self.lb), size)
        self.recall = 0
        for truth in self.img_truth:
            self.recall += float(np.sum(bd & truth)) / float(np.sum(truth))
        self.recall /= len(self.img_truth)

    def set_boundary_precision(self, size=5):
        # the reference ignores ``size`` here and hard-codes 5 (metrics.py:93)
        bd = find_boundaries(self.lb)
        self.precision = 0
        global_score = float(np.sum(bd))
        for truth in self.img_truth:
            self.precision += float(np.sum(bd & _dilate(truth, 5))) / global_score
        self.precision /= len(self.img_truth)

    def set_fmeasure(self):
        """F = 2PR/(P+R), 0 when P+R = 0. Not in the reference."""
        s = self.recall + self.precision
        self.fmeasure = 0.0 if s == 0 else 2.0 * self.precision * self.recall / s

    # ---- undersegmentation (metrics.py:102-146), contingency table by bincount
    def set_undersegmentation(self):
        self.undersegmentation = 0.
        self.undersegmentationNP = 0.
        n = self.nx * self.ny
        for truth in self.segments_truth:
            truth = np.asarray(truth).astype(np.int64)
            n_labels = int(np.max(truth) + 1)
            hist = np.bincount((self.lb.astype(np.int64) * n_labels + truth).ravel(),
                               minlength=int(self.n_segments) * n_labels)
            hist = hist.reshape(int(self.n_segments), n_labels).astype(np.float64)
            area = hist.sum(axis=1)
            self.undersegmentation += float(np.sum(area - hist.max(axis=1))) / n
            self.undersegmentationNP += float(np.sum(np.minimum(hist, area[:, None] - hist))) / n
        self.undersegmentation /= len(self.segments_truth)
        self.undersegmentationNP /= len(self.segments_truth)

    # ---- geometry (metrics.py:152-201)
    def set_density(self):
        self.density = np.sum(find_boundaries(self.lb)) / float(self.nx * self.ny)

    def perimeter(self):
        lb = self.lb
        edge = np.zeros(lb.shape, bool)
        edge[0, :] = edge[-1, :] = True
        edge[:, 0] = edge[:, -1] = True
        inner = np.zeros(lb.shape, bool)
        c = lb[1:-1, 1:-1]
        inner[1:-1, 1:-1] = (lb[:-2, 1:-1] != c) | (lb[2:, 1:-1] != c) | (lb[1:-1, :-2] != c) | (lb[1:-1, 2:] != c)
        self.perimeters = np.bincount(lb[edge | inner].ravel(), minlength=int(self.n_segments)).astype(np.float64)

    def set_compactness(self):
        self.perimeter()
        self.compactness = 0
        max_area = float(self.nx * self.ny)
        areas = np.bincount(self.lb.ravel(), minlength=int(self.n_segments))
        for i in range(int(self.n_segments)):
            area = areas[i]
            perimeter = self.perimeters[i]
            ratio = area / max_area
            if perimeter > 0:
                self.compactness += 4 * pi * ratio * area / pow(perimeter, 2)

    # ---- orchestration (metrics.py:208-255)
    def set_metrics(self):
        self.set_boundary_recall()
        self.set_boundary_precision()
        self.set_fmeasure()
        self.set_density()
        self.set_undersegmentation()
        self.set_compactness()

    def display_metrics(self):
        print("Regions: " + str(self.n_segments) +
              " Recall: " + str(self.recall) +
              " Precision: " + str(self.precision) +
              " F-measure: " + str(self.fmeasure) +
              " Undersegmentation: " + str(self.undersegmentation) +
              " Undersegmentation (NP) " + str(self.undersegmentationNP) +
              " Compactness " + str(self.compactness) +
              " Density " + str(self.density))

    def get_metrics(self):
        return {"regions": self.n_segments, "recall": self.recall, "precision": self.precision,
                "fmeasure": self.fmeasure, "underseg": self.undersegmentation,
                "undersegNP": self.undersegmentationNP, "compactness": self.compactness,
                "density": self.density}


def boundary_scores(labels, segments_truth) -> dict:
    """Recall / precision / F of one label map (the part of the metric BASELINE.json names)."""
    m = metrics(None, labels, segments_truth)
    m.set_boundary_recall()
    m.set_boundary_precision()
    m.set_fmeasure()
    return {"recall": m.recall, "precision": m.precision, "fmeasure": m.fmeasure}
