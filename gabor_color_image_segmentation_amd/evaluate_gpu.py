"""Boundary recall / precision / F on the GPU.

Same quantities as ``evaluate.metrics.set_boundary_recall/precision`` — i.e. as
/root/reference/BSD_metrics/metrics.py:58-96 — with the stencils and the masked sums done by
``gcs_boundary_counts`` (integer counts) and the reference's float divisions / per-annotator
mean done here in the reference's order, so the results are identical floats. Lets large
batches be scored without a device->host round trip of the label maps.
"""
from __future__ import annotations

import numpy as np

from . import _lib


def boundary_counts_device(labels, truths):
    """labels: (H,W) int32 device tensor; truths: (A,H,W) int16/uint16-bits device tensor.
    Returns the uint64 counts [1 + 3A] as a host numpy array."""
    import torch
    lib = _lib.load()
    if labels.dtype != torch.int32 or labels.dim() != 2:
        raise ValueError("labels must be an (H,W) int32 tensor")
    if truths.dim() != 3 or truths.shape[1:] != labels.shape or truths.element_size() != 2:
        raise ValueError("truths must be an (A,H,W) 16-bit tensor matching labels")
    a, h, w = truths.shape
    labels, truths = labels.contiguous(), truths.contiguous()
    scratch = torch.empty(lib.gcs_boundary_scratch_bytes(a, h, w), dtype=torch.uint8, device=labels.device)
    counts = torch.empty(1 + 3 * a, dtype=torch.int64, device=labels.device)
    _lib.check(lib.gcs_boundary_counts(labels.data_ptr(), truths.data_ptr(), a, h, w, scratch.data_ptr(),
                                       counts.data_ptr(), torch.cuda.current_stream(labels.device).cuda_stream),
               "gcs_boundary_counts")
    return counts.cpu().numpy().astype(np.uint64)


def scores_from_counts(counts) -> dict:
    """metrics.py:69-74 and :88-96 arithmetic on the integer counts (plain Python floats)."""
    a = (len(counts) - 1) // 3
    recall = 0
    precision = 0
    global_score = float(counts[0])
    for i in range(a):
        recall += float(counts[1 + 3 * i]) / float(counts[2 + 3 * i])        # ZeroDivisionError as metrics.py:72
        precision += float(counts[3 + 3 * i]) / global_score                 # ZeroDivisionError as metrics.py:94
    recall /= a
    precision /= a
    s = recall + precision
    return {"recall": recall, "precision": precision,
            "fmeasure": 0.0 if s == 0 else 2.0 * precision * recall / s}


def boundary_scores_device(labels, segments_truth) -> dict:
    """labels: (H,W) int32 device tensor (e.g. a row of Segmenter.segment_device);
    segments_truth: list of (H,W) integer arrays (groundtruth.get_segment_from_filename)."""
    import torch
    if len(segments_truth) == 0:
        raise ZeroDivisionError("no annotator maps (metrics.py:74 divides by len(img_truth))")
    t = np.stack([np.asarray(s).astype(np.uint16) for s in segments_truth]).view(np.int16)
    truths = torch.from_numpy(np.ascontiguousarray(t)).to(labels.device)
    return scores_from_counts(boundary_counts_device(labels, truths))
