"""CPU ORACLE — TEST INFRASTRUCTURE ONLY. Never imported by the product path.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module, and only as the checker / the timed CPU baseline.

What it restates
----------------
* Gabor bank + features + k-means: **parity unpinned**. The reference tree holds no
  Gabor or k-means code (the only segmenter call is scikit-image SLIC
  at /root/reference/BSD_metrics/script.py:30), so these functions restate SPEC.md
  §2-§4 in plain NumPy + scipy.ndimage. This is "the reference NumPy/scipy CPU path"
  BASELINE.json's north_star wants timed beside the GPU.
* Boundary recall / precision: restates /root/reference/BSD_metrics/metrics.py:25-51
  (constructor: thick boundaries of each annotator map), :58-74 (recall), :77-96
  (precision) on scipy.ndimage, because scikit-image is absent from the torch
  interpreter. **Pinned** by tests/golden/scoring_golden.npz, produced by running the
  reference's own ``metrics`` class (tests/golden/make_scoring_golden.py).
"""
from __future__ import annotations

import math

import numpy as np
from scipy import ndimage as ndi


_CROSS3 = ndi.generate_binary_structure(2, 1)


# --------------------------------------------------------------------------- bank
def bank(n_scales=4, n_orient=6, ksize=13, f_max=0.4, ratio=math.sqrt(2.0), bandwidth=1.0):
    """SPEC.md §2. Returns (tapq int64 [F,2,ks,ks], shift)."""
    r = (ksize - 1) // 2
    ax = np.arange(-r, r + 1, dtype=np.float64)
    dy = ax[:, None] * np.ones((1, ksize))
    dx = np.ones((ksize, 1)) * ax[None, :]
    kappa = math.sqrt(math.log(2.0) / 2.0) / math.pi * (2.0 ** bandwidth + 1.0) / (2.0 ** bandwidth - 1.0)
    taps = []
    for s in range(n_scales):
        freq = f_max / ratio ** s * 2.0 ** (s // 2)      # f_base on pyramid level s // 2
        sigma = kappa / freq
        env = np.exp(-(dx * dx + dy * dy) / (2.0 * sigma * sigma))
        env = env / env.sum()
        for o in range(n_orient):
            theta = o * math.pi / n_orient
            phase = 2.0 * math.pi * freq * (dx * math.cos(theta) + dy * math.sin(theta))
            taps.append(np.stack([env * np.cos(phase), env * np.sin(phase)]))
    taps = np.stack(taps)
    e = min(15, int(math.floor(math.log2(32639.0 / np.abs(taps).max()))))
    tapq = np.rint(taps * 2.0 ** e).astype(np.int64)
    return tapq, e - 7


# ----------------------------------------------------------------------- features
def isqrt_array(n: np.ndarray) -> np.ndarray:
    """Exact floor(sqrt(n)) for non-negative int64 arrays below 2**52."""
    q = np.floor(np.sqrt(n.astype(np.float64))).astype(np.int64)
    q -= (q * q > n)
    q += ((q + 1) * (q + 1) <= n)
    return q


def pyramid(img: np.ndarray, n_levels: int):
    """SPEC.md §3 pyramid: I_0 = img, I_{L+1} = 2x2 block mean (round half up) of I_L edge-replicated to even size."""
    levels = [np.asarray(img)]
    for _ in range(1, n_levels):
        a = levels[-1].astype(np.int64)
        a = np.pad(a, ((0, a.shape[0] & 1), (0, a.shape[1] & 1), (0, 0)), mode="edge")
        levels.append(((a[0::2, 0::2] + a[0::2, 1::2] + a[1::2, 0::2] + a[1::2, 1::2] + 2) >> 2).astype(np.uint8))
    return levels


def level_of(f: int, n_orient: int) -> int:
    return (f // n_orient) // 2


def gabor_features_levels(img: np.ndarray, tapq: np.ndarray, shift: int, n_orient: int):
    """SPEC.md §3 responses at their own resolution: list over levels of (3F, H_L, W_L) uint16 arrays in which
    only the rows d = c*F + f of that level's filters are filled (the others stay zero)."""
    assert img.dtype == np.uint8 and img.ndim == 3 and img.shape[2] == 3
    nf = tapq.shape[0]
    n_levels = level_of(nf - 1, n_orient) + 1
    out = []
    for lv, im in enumerate(pyramid(img, n_levels)):
        g = np.zeros((3 * nf,) + im.shape[:2], np.uint16)
        for c in range(3):
            chan = im[:, :, c].astype(np.int64)
            for f in range(nf):
                if level_of(f, n_orient) != lv:
                    continue
                v_re = ndi.correlate(chan, tapq[f, 0], mode='reflect')
                v_im = ndi.correlate(chan, tapq[f, 1], mode='reflect')
                a_re = v_re >> shift
                a_im = v_im >> shift
                g[c * nf + f] = isqrt_array(a_re * a_re + a_im * a_im)
        out.append(g)
    return out


def gabor_features(img: np.ndarray, tapq: np.ndarray, shift: int, n_orient: int) -> np.ndarray:
    """SPEC.md §3. img (H,W,3) uint8 -> canonical feats (3F,H,W) uint16, d = c*F + f: level-L responses
    replicated over their 2^L x 2^L blocks."""
    nf = tapq.shape[0]
    h, w = img.shape[:2]
    out = np.empty((3 * nf, h, w), np.uint16)
    for lv, g in enumerate(gabor_features_levels(img, tapq, shift, n_orient)):
        rows = [c * nf + f for c in range(3) for f in range(nf) if level_of(f, n_orient) == lv]
        up = g[rows].repeat(1 << lv, axis=1).repeat(1 << lv, axis=2)
        out[rows] = up[:, :h, :w]
    return out


# ------------------------------------------------------------------------ k-means
def kmeans_init(x0: np.ndarray, k: int) -> np.ndarray:
    """SPEC.md §4 init. x0 (P,D) integer features of ONE image -> (k,D) int64."""
    p = x0.shape[0]
    idx = [((2 * j + 1) * p) // (2 * k) for j in range(k)]
    return x0[idx].astype(np.int64)


def kmeans_assign(x: np.ndarray, c: np.ndarray) -> np.ndarray:
    """argmin_j sum_d (x_pd - c_jd)^2, exact int64, ties -> lowest j."""
    x = x.astype(np.int64)
    best = None
    lab = np.zeros(x.shape[0], np.int32)
    for j in range(c.shape[0]):
        diff = x - c[j][None, :]
        dist = (diff * diff).sum(axis=1)
        if best is None:
            best = dist
        else:
            m = dist < best
            lab[m] = j
            best = np.where(m, dist, best)
    return lab


def kmeans_update(x: np.ndarray, lab: np.ndarray, c: np.ndarray):
    """Returns (new centroids, counts, sums); empty clusters keep their centroid."""
    k, d = c.shape
    x = x.astype(np.int64)
    cnt = np.bincount(lab, minlength=k).astype(np.int64)
    sums = np.zeros((k, d), np.int64)
    for j in range(k):
        sums[j] = x[lab == j].sum(axis=0)
    new = c.copy()
    nz = cnt > 0
    new[nz] = (2 * sums[nz] + cnt[nz, None]) // (2 * cnt[nz, None])
    return new, cnt, sums


def kmeans(x: np.ndarray, k: int, n_iter: int, init_from: np.ndarray | None = None):
    """SPEC.md §4 schedule on a (P,D) matrix. Returns (labels int32, centroids)."""
    c = kmeans_init(x if init_from is None else init_from, k)
    lab = None
    for t in range(n_iter):
        lab = kmeans_assign(x, c)
        if t < n_iter - 1:
            c, _, _ = kmeans_update(x, lab, c)
    return lab, c


# ------------------------------------------------------------------------ segment
def segment(img, n_scales=4, n_orient=6, k=8, n_iter=10, ksize=13, f_max=0.4,
            ratio=math.sqrt(2.0), bandwidth=1.0, return_all=False):
    """segment(image) -> (H,W) int32 label map (per-image codebook)."""
    tapq, shift = bank(n_scales, n_orient, ksize, f_max, ratio, bandwidth)
    feats = gabor_features(img, tapq, shift, n_orient)
    h, w = img.shape[:2]
    x = feats.reshape(feats.shape[0], -1).T
    lab, c = kmeans(x, k, n_iter)
    lab = lab.reshape(h, w).astype(np.int32)
    return (lab, feats, c) if return_all else lab


def segment_batch(imgs, mode="per_image", **kw):
    """(B,H,W,3) -> (B,H,W) int32. mode 'global' = one codebook for the batch."""
    k = kw.get("k", 8)
    n_iter = kw.get("n_iter", 10)
    bank_kw = {a: kw[a] for a in ("n_scales", "n_orient", "ksize", "f_max", "ratio", "bandwidth") if a in kw}
    if mode == "per_image":
        return np.stack([segment(im, k=k, n_iter=n_iter, **bank_kw) for im in imgs])
    tapq, shift = bank(**bank_kw)
    b, h, w = imgs.shape[:3]
    n_orient = bank_kw.get("n_orient", 6)
    xs = [gabor_features(im, tapq, shift, n_orient).reshape(3 * tapq.shape[0], -1).T for im in imgs]
    x = np.concatenate(xs)
    lab, _ = kmeans(x, k, n_iter, init_from=xs[0])
    return lab.reshape(b, h, w).astype(np.int32)


# -------------------------------------------------------------- connected regions
def connected_regions(lab: np.ndarray) -> np.ndarray:
    """SPEC.md §7: 4-connected components of equal labels, ids in raster order of first pixel."""
    lab = np.asarray(lab)
    comp = np.zeros(lab.shape, np.int64)
    n = 0
    for v in np.unique(lab):
        c, m = ndi.label(lab == v, structure=_CROSS3)
        comp[c > 0] = c[c > 0] + n
        n += m
    # renumber by first occurrence in raster order
    flat = comp.ravel()
    _, first = np.unique(flat, return_index=True)
    order = np.argsort(first)                       # component ids (1-based, sorted) by first pixel
    remap = np.empty(n + 1, np.int64)
    remap[np.unique(flat)[order]] = np.arange(n)
    return remap[flat].reshape(lab.shape).astype(np.int32)


# ------------------------------------------------------------------------ scoring
_SQ5 = np.ones((5, 5), bool)


def find_boundaries_thick(lab: np.ndarray) -> np.ndarray:
    """skimage find_boundaries(mode='thick', connectivity=1) as used at metrics.py:49,69,88."""
    lab = np.asarray(lab)
    return ndi.grey_dilation(lab, footprint=_CROSS3) != ndi.grey_erosion(lab, footprint=_CROSS3)


def boundary_recall_precision(lab: np.ndarray, truths) -> tuple[float, float]:
    """metrics.py:58-74 (recall, 5x5 dilation of the label boundaries) and
    metrics.py:77-96 (precision, 5x5 dilation of each annotator's boundaries)."""
    lab = np.asarray(lab).astype('int')
    bd = find_boundaries_thick(lab)
    bd_dil = ndi.binary_dilation(bd, structure=_SQ5)
    tb = [find_boundaries_thick(t) for t in truths]
    recall = sum(float(np.sum(bd_dil & t)) / float(np.sum(t)) for t in tb) / len(tb)
    g = float(np.sum(bd))
    precision = sum(float(np.sum(bd & ndi.binary_dilation(t, structure=_SQ5))) / g for t in tb) / len(tb)
    return recall, precision


def fmeasure(recall: float, precision: float) -> float:
    return 0.0 if recall + precision == 0 else 2.0 * precision * recall / (precision + recall)
