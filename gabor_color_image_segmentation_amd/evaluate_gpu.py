"""Scoring on the GPU: boundary recall / precision / F and the region metrics
(undersegmentation, compactness, density: §8f rank 2).

Same quantities as ``evaluate.metrics.set_boundary_recall/precision`` — i.e. as
/root/reference/BSD_metrics/metrics.py:58-96 — with the stencils and the masked sums done by
``gcs_boundary_counts`` (integer counts) and the reference's float divisions / per-annotator
mean done here in the reference's order, so the results are identical floats. Lets large
batches be scored without a device->host round trip of the label maps.
"""
from __future__ import annotations

import numpy as np

from math import pi

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


def _truth_stack(segments_truth, device):
    import torch
    t = np.stack([np.asarray(s).astype(np.uint16) for s in segments_truth])
    return t, torch.from_numpy(np.ascontiguousarray(t.view(np.int16))).to(device)


def region_counts_device(labels, truths, n_segments, n_truth_labels):
    """labels: (H,W) int32 device tensor with values < n_segments; truths: (A,H,W) 16-bit device tensor with
    values < n_truth_labels. Returns host arrays (hist [A,n_segments,n_truth_labels], area, perimeters)."""
    import torch
    lib = _lib.load()
    if labels.dtype != torch.int32 or labels.dim() != 2:
        raise ValueError("labels must be an (H,W) int32 tensor")
    if truths.dim() != 3 or truths.shape[1:] != labels.shape or truths.element_size() != 2:
        raise ValueError("truths must be an (A,H,W) 16-bit tensor matching labels")
    a, h, w = truths.shape
    labels, truths = labels.contiguous(), truths.contiguous()
    hist = torch.empty((a, n_segments, n_truth_labels), dtype=torch.int32, device=labels.device)
    area = torch.empty(n_segments, dtype=torch.int32, device=labels.device)
    perim = torch.empty(n_segments, dtype=torch.int32, device=labels.device)
    _lib.check(lib.gcs_region_counts(labels.data_ptr(), truths.data_ptr(), a, h, w, int(n_segments),
                                     int(n_truth_labels), hist.data_ptr(), area.data_ptr(), perim.data_ptr(),
                                     torch.cuda.current_stream(labels.device).cuda_stream), "gcs_region_counts")
    return hist.cpu().numpy(), area.cpu().numpy(), perim.cpu().numpy()


def region_scores_from_counts(hist, area, perim, n_truth, nx, ny) -> dict:
    """metrics.py:128-146 and :188-201 arithmetic on the integer tables, giving the reference's floats.
    ``n_truth[a]`` = max(truth_a) + 1 (metrics.py:116): only those columns of annotator a exist there.

    Where the reference adds up integer-valued terms (all far below 2^53: every order gives the same float64) the loops over
    segments are array operations here; where it adds up fractions (per annotator, per segment of the compactness) the order of
    the reference's loops is kept, term by term. (The per-segment Python loops of round 3 cost 0.24 ms per image - more than
    the kernels that produce the tables.)"""
    area_f = area.astype(np.float64)
    under = 0.
    under_np = 0.
    for a in range(hist.shape[0]):
        h = hist[a][:, :int(n_truth[a])].astype(np.float64)
        u = float(np.sum(area_f - h.max(axis=1)))                # metrics.py:129-130: integer-valued terms, exact in any order
        u /= nx * ny
        under += u
        unp = float(np.sum(np.minimum(h, h.sum(axis=1)[:, None] - h)))   # metrics.py:137-139: integers, any order
        unp /= nx * ny
        under_np += unp
    under /= hist.shape[0]
    under_np /= hist.shape[0]
    # metrics.py:194-201: compactness += 4 pi (a / max_area) a / perimeter^2 over the segments with a perimeter, in index order.
    # Element by element the same float64 operations in the same order as the reference's scalar expression (perimeter^2 is
    # an integer below 2^53: exact however it is computed); the sum itself stays a left-to-right loop.
    max_area = float(nx * ny)
    a64 = area.astype(np.int64)
    per = perim.astype(np.float64)
    has = per > 0
    terms = 4 * pi * (a64[has] / max_area) * a64[has] / (per[has] * per[has])
    compactness = 0
    for t in terms:
        compactness += t
    return {"underseg": float(under), "undersegNP": float(under_np), "compactness": float(compactness)}


def all_scores_device(labels, segments_truth) -> dict:
    """Every number of ``evaluate.metrics.get_metrics()`` (= metrics.py:246-255 plus F) for a device label map:
    stencils, masked sums and histograms on the GPU, the reference's float arithmetic on the host."""
    if len(segments_truth) == 0:
        raise ZeroDivisionError("no annotator maps (metrics.py:74 divides by len(img_truth))")
    t, truths = _truth_stack(segments_truth, labels.device)
    nx, ny = labels.shape
    n_segments = int(labels.max().item()) + 1                    # metrics.py:51
    n_truth = [int(s.max()) + 1 for s in t]                      # metrics.py:116
    counts = boundary_counts_device(labels, truths)
    out = {"regions": n_segments}
    out.update(scores_from_counts(counts))
    hist, area, perim = region_counts_device(labels, truths, n_segments, max(n_truth))
    out.update(region_scores_from_counts(hist, area, perim, n_truth, nx, ny))
    out["density"] = float(counts[0]) / float(nx * ny)           # metrics.py:157
    return out


def all_scores_batch_device(labels, truth, first=None, img_of=None, n_truth=None, n_segments=None) -> list:
    """Every number of ``evaluate.metrics.get_metrics()`` for a whole batch of device label maps in THREE launches
    (boundary maps, boundary counts, region tables) and one device-to-host copy per table, instead of a scoring call
    and two host round trips per image (metrics.py:58-201 loops over images in script.py:22).

    labels: (B,H,W) int32 device tensor; truth / first / img_of / n_truth: ``PackedTruth.stack(ids)`` (or the same
    layout built by hand): all annotator maps of image 0, then of image 1, ...; n_segments: max label + 1 over the
    batch (default: read from the labels; metrics.py:51 per image is the image's own max + 1, applied below)."""
    import torch
    if isinstance(truth, DeviceTruth):                       # resident ground truth (round 5): nothing of it is uploaded or re-derived
        return all_scores_batch_resident(labels, truth, n_segments)
    lib = _lib.load()
    if labels.dtype != torch.int32 or labels.dim() != 3:
        raise ValueError("labels must be a (B,H,W) int32 tensor")
    b, h, w = labels.shape
    truth = np.ascontiguousarray(truth, np.uint16)
    t = truth.shape[0]
    if truth.shape[1:] != (h, w) or len(first) != b + 1 or int(first[-1]) != t or len(img_of) != t:
        raise ValueError("truth stack does not match the label batch")
    if any(int(first[i + 1]) == int(first[i]) for i in range(b)):
        raise ZeroDivisionError("an image has no annotator maps (metrics.py:74 divides by len(img_truth))")
    dev = labels.device
    labels = labels.contiguous()
    seg_max = labels.reshape(b, -1).max(dim=1).values                   # per-image max label, one small copy below
    truth_d = torch.from_numpy(truth.view(np.int16)).to(dev)
    first_d = torch.from_numpy(np.ascontiguousarray(first, np.int32)).to(dev)
    img_of_d = torch.from_numpy(np.ascontiguousarray(img_of, np.int32)).to(dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    scratch = torch.empty(lib.gcs_boundary_batch_scratch_bytes(b, t, h, w), dtype=torch.uint8, device=dev)
    counts = torch.empty(b + 3 * t, dtype=torch.int64, device=dev)
    _lib.check(lib.gcs_boundary_counts_batch(labels.data_ptr(), truth_d.data_ptr(), img_of_d.data_ptr(), b, t, h, w,
                                             scratch.data_ptr(), counts.data_ptr(), stream), "gcs_boundary_counts_batch")
    seg_max = seg_max.cpu().numpy()
    n_seg = int(n_segments) if n_segments is not None else int(seg_max.max()) + 1
    stride = int(max(n_truth))
    a_max = int(max(int(first[i + 1]) - int(first[i]) for i in range(b)))
    hist = torch.empty((t, n_seg, stride), dtype=torch.int32, device=dev)
    area = torch.empty((b, n_seg), dtype=torch.int32, device=dev)
    perim = torch.empty((b, n_seg), dtype=torch.int32, device=dev)
    _lib.check(lib.gcs_region_counts_batch(labels.data_ptr(), truth_d.data_ptr(), first_d.data_ptr(), b, t, a_max, h, w,
                                           n_seg, stride, hist.data_ptr(), area.data_ptr(), perim.data_ptr(), stream),
               "gcs_region_counts_batch")
    counts = counts.cpu().numpy().astype(np.uint64)
    hist, area, perim = hist.cpu().numpy(), area.cpu().numpy(), perim.cpu().numpy()
    out = []
    for i in range(b):
        t0, t1 = int(first[i]), int(first[i + 1])
        c = np.concatenate([counts[i:i + 1], counts[b + 3 * t0:b + 3 * t1]])   # the single-image layout [1 + 3A]
        n_i = int(seg_max[i]) + 1                                                # metrics.py:51
        res = {"regions": n_i}
        res.update(scores_from_counts(c))
        res.update(region_scores_from_counts(hist[t0:t1, :n_i], area[i, :n_i], perim[i, :n_i], n_truth[t0:t1], h, w))
        res["density"] = float(c[0]) / float(h * w)
        out.append(res)
    return out


# ------------------------------------------------------------------------------------------------------------------------
# Resident ground truth (round 5). metrics.py:48-49 derives find_boundaries(truth) anew for every image it scores and
# groundtruth.py:44-48 rescans the split directories per id; round 4's batched scorer still uploaded the annotator maps (40 MB per
# 24 images) and re-derived their boundary / dilated planes on every call - 67 % of a segment + score loop. The maps are
# constants of the data set: a DeviceTruth holds them on the device once, as uint8 maps (region tables) and as bit planes of
# bd(T) and dil5(bd(T)) with the counts sum bd(T) (gcs_truth_prepare), and a scoring call touches nothing else of them.

class DeviceTruth:
    """The annotator maps of a list of equally shaped images, resident on a device in the form the scorer consumes.

    Build with ``DeviceTruth(truth, first, img_of, n_truth, device)`` from a ``PackedTruth.stack(ids)`` tuple, or
    ``PackedTruth.to_device(ids, device)``. Pass it to ``all_scores_batch_device(labels, device_truth)``."""

    def __init__(self, truth, first, img_of, n_truth, device="cuda"):
        import torch
        lib = _lib.load()
        truth = np.ascontiguousarray(truth, np.uint16)
        self.t, self.h, self.w = (int(x) for x in truth.shape)
        self.first = np.ascontiguousarray(first, np.int32)
        self.img_of = np.ascontiguousarray(img_of, np.int32)
        self.b = len(self.first) - 1
        if int(self.first[-1]) != self.t or len(self.img_of) != self.t:
            raise ValueError("first / img_of do not describe the truth stack")
        if np.any(np.diff(self.first) <= 0):
            raise ZeroDivisionError("an image has no annotator maps (metrics.py:74 divides by len(img_truth))")
        self.n_truth = np.asarray(n_truth, np.int64)
        self.stride = int(self.n_truth.max())
        self.a_max = int(np.diff(self.first).max())
        dev = torch.device(device)
        if dev.type == "cuda" and dev.index is None:           # 'cuda' -> the current device, so that it compares equal to a tensor's
            dev = torch.device("cuda", torch.cuda.current_device())
        self.device = dev
        with torch.cuda.device(dev):
            t16 = torch.from_numpy(truth.view(np.int16)).to(dev)
            self.planes = torch.empty(lib.gcs_bit_planes_bytes(self.t, self.h, self.w), dtype=torch.uint8, device=dev)
            self.bd_counts = torch.empty(self.t, dtype=torch.int64, device=dev)
            self.u8 = self.stride <= 256                       # BSD500: the largest annotator label is 208
            self.maps = torch.empty((self.t, self.h, self.w), dtype=torch.uint8, device=dev) if self.u8 else t16
            _lib.check(lib.gcs_truth_prepare(t16.data_ptr(), self.t, self.h, self.w, self.planes.data_ptr(),
                                             self.bd_counts.data_ptr(), self.maps.data_ptr() if self.u8 else None,
                                             torch.cuda.current_stream(dev).cuda_stream), "gcs_truth_prepare")
            self.first_d = torch.from_numpy(self.first).to(dev)
            self.img_of_d = torch.from_numpy(self.img_of).to(dev)
            torch.cuda.current_stream(dev).synchronize()       # t16 may go (uint8 case): the kernels that read it are done
        self._out = None                                       # (capacity, device result block, pinned host block, views, ...)
        self._scratch = None

    def _buffers(self, n_seg):
        """ONE device block for everything a call returns (counts | under | under_np | seg_max | area | perim) and its pinned
        mirror: one device-to-host copy of a few KB and one synchronisation per call instead of five; the contingency tables
        stay on the device (gcs_region_reduce takes the two sums metrics.py:128-140 needs out of them).

        ONE entry per DeviceTruth, sized to a CAPACITY of segments (the next power of two, at least 8): a data-set loop over
        connected-region maps, whose label count differs from batch to batch, reuses it and reallocates - dropping the old
        blocks - only when a batch needs more (the kernels take the capacity as their table stride; segments that do not occur
        have area 0 and perimeter 0 and add nothing to any score). Returns (capacity, device block, pinned block, offsets,
        bit-plane scratch, contingency tables)."""
        import torch
        cap = 8                                                 # (k-means maps of the default k stay at their exact size)
        while cap < n_seg:
            cap *= 2
        if self._scratch is None:                               # the label maps' bit planes: does not depend on n_seg
            self._scratch = torch.empty(_lib.load().gcs_bit_planes_bytes(self.b, self.h, self.w), dtype=torch.uint8,
                                        device=self.device)
        ent = self._out
        if ent is None or ent[0] < cap:
            self._out = ent = None                              # the old blocks go back to the allocator first
            b, t = self.b, self.t
            sizes = [("counts", (b + 3 * t) * 8), ("under", t * 8), ("under_np", t * 8), ("seg_max", b * 4),
                     ("area", b * cap * 4), ("perim", b * cap * 4)]
            offs, o = {}, 0
            for name, nbytes in sizes:
                offs[name] = (o, nbytes)
                o += (nbytes + 15) // 16 * 16
            dev_blk = torch.empty(o, dtype=torch.uint8, device=self.device)
            host_blk = torch.empty(o, dtype=torch.uint8, pin_memory=True)
            hist = torch.empty(t * cap * self.stride, dtype=torch.int32, device=self.device)       # never leaves the device
            ent = self._out = (cap, dev_blk, host_blk, offs, self._scratch, hist)
        return ent


def _region_scores_batch(under, under_np, area, perim, first, nx, ny):
    """``region_scores_from_counts`` for a whole batch: under / under_np uint64 [T] = the integer sums of gcs_region_reduce
    (metrics.py:129-130, :137-139; exact in any order), area / perim [B][n_seg]. The sums of fractions keep the reference's
    order, term by term (metrics.py:131, :140, :194-201)."""
    b = len(first) - 1
    u_t = under.astype(np.float64) / (nx * ny)
    unp_t = under_np.astype(np.float64) / (nx * ny)
    max_area = float(nx * ny)
    a64 = area.astype(np.int64)
    per = perim.astype(np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        terms = 4 * pi * (a64 / max_area) * a64 / (per * per)                    # metrics.py:199, element by element
    out = []
    u_t, unp_t, first = u_t.tolist(), unp_t.tolist(), np.asarray(first).tolist()  # Python floats: the same IEEE doubles, no
    terms, has = terms.tolist(), (per > 0).tolist()                              # NumPy scalar boxing per addition
    for i in range(b):
        t0, t1 = first[i], first[i + 1]
        under_i = 0.
        under_np_i = 0.
        for t in range(t0, t1):                                                  # metrics.py:131,140: += in annotator order
            under_i += u_t[t]
            under_np_i += unp_t[t]
        compactness = 0
        for term, ok in zip(terms[i], has[i]):                                   # metrics.py:194-201: index order
            if ok:
                compactness += term
        out.append({"underseg": under_i / (t1 - t0), "undersegNP": under_np_i / (t1 - t0), "compactness": float(compactness)})
    return out


class _PendingScores:
    """Scores of one batch on their way: kernels and the result copy are enqueued, ``result()`` waits for the copy and does the
    reference's float arithmetic. Submitting the next batch before collecting this one lets its kernels run under that
    arithmetic (``all_scores_batch_resident`` = submit + result)."""

    def __init__(self, truth, b, h, w, n_seg, host_blk, offs, event, cap=None):
        self._a = (truth, b, h, w, n_seg, host_blk, offs, event, cap or n_seg)

    def result(self) -> list:
        truth, b, h, w, n_seg, host_blk, offs, event, cap = self._a
        event.synchronize()
        raw = host_blk.numpy()
        view = lambda k, dt: raw[offs[k][0]:offs[k][0] + offs[k][1]].view(dt)
        counts = view("counts", np.uint64)
        seg_max = view("seg_max", np.int32)
        if int(seg_max.max()) >= n_seg:
            raise ValueError(f"a label map holds label {int(seg_max.max())} but n_segments = {n_seg}")
        area = view("area", np.int32).reshape(b, cap)[:, :n_seg]     # tables are laid out at the capacity; columns >= n_seg are zero
        perim = view("perim", np.int32).reshape(b, cap)[:, :n_seg]
        reg = _region_scores_batch(view("under", np.uint64), view("under_np", np.uint64), area, perim, truth.first, h, w)
        out = []
        cf = counts.astype(np.float64).tolist()                      # counts < 2^53: exact; Python floats from here on
        first = truth.first.tolist()
        for i in range(b):
            t0, t1 = first[i], first[i + 1]
            g = cf[i]
            recall = 0
            precision = 0
            for t in range(t0, t1):                                  # metrics.py:69-74, :88-96 in annotator order
                recall += cf[b + 3 * t] / cf[b + 3 * t + 1]          # ZeroDivisionError as metrics.py:72
                precision += cf[b + 3 * t + 2] / g                   # ZeroDivisionError as metrics.py:94
            recall /= t1 - t0
            precision /= t1 - t0
            sm = recall + precision
            res = {"regions": int(seg_max[i]) + 1, "recall": recall, "precision": precision,
                   "fmeasure": 0.0 if sm == 0 else 2.0 * precision * recall / sm}
            res.update(reg[i])
            res["density"] = g / float(h * w)
            out.append(res)
        return out


def submit_scores_batch_resident(labels, truth: DeviceTruth, n_segments=None) -> _PendingScores:
    """Enqueue the scoring of a (B,H,W) int32 device label batch against resident ground truth; ``.result()`` returns what
    ``all_scores_batch_device`` returns. One result block per DeviceTruth: collect a submission before submitting the
    next batch against the SAME DeviceTruth."""
    import torch
    lib = _lib.load()
    if labels.dtype != torch.int32 or labels.dim() != 3:
        raise ValueError("labels must be a (B,H,W) int32 tensor")
    b, h, w = labels.shape
    if (b, h, w) != (truth.b, truth.h, truth.w) or labels.device != truth.device:
        raise ValueError("label batch does not match the resident truth (images, shape or device)")
    labels = labels.contiguous()
    n_seg = int(n_segments) if n_segments is not None else int(labels.max().item()) + 1
    cap, dev_blk, host_blk, offs, scratch, hist_d = truth._buffers(n_seg)
    base = dev_blk.data_ptr()
    ptr = {k: base + o for k, (o, _) in offs.items()}
    with torch.cuda.device(truth.device):
        stream = torch.cuda.current_stream(truth.device)
        _lib.check(lib.gcs_score_batch_resident(labels.data_ptr(), truth.planes.data_ptr(), truth.bd_counts.data_ptr(),
                                                truth.maps.data_ptr(), 1 if truth.u8 else 0, truth.first_d.data_ptr(),
                                                truth.img_of_d.data_ptr(), b, truth.t, truth.a_max, h, w, cap, truth.stride,
                                                scratch.data_ptr(), hist_d.data_ptr(), ptr["counts"], ptr["seg_max"], ptr["area"],
                                                ptr["perim"], ptr["under"], ptr["under_np"], stream.cuda_stream),
                   "gcs_score_batch_resident")
        host_blk.copy_(dev_blk, non_blocking=True)
        event = torch.cuda.Event()
        event.record(stream)
    return _PendingScores(truth, b, h, w, n_seg, host_blk, offs, event, cap)


def all_scores_batch_resident(labels, truth: DeviceTruth, n_segments=None) -> list:
    """``all_scores_batch_device`` on resident ground truth: kernels on bit planes and uint8 maps, the tables reduced on the
    device, ONE device-to-host copy of a few KB. ``n_segments``: an upper bound of max label + 1 over the batch (the
    Segmenter's k); None reads it from the labels (one more synchronisation)."""
    return submit_scores_batch_resident(labels, truth, n_segments).result()
