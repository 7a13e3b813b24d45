#!/usr/bin/env python3
"""Design study: how many 256-pixel k-means tiles could an EXACT bound-based pass skip?

Tile-level Hamerly test. When a tile is processed in pass t0, every pixel's slack s_p = sqrt(d2_second) - sqrt(d2_best)
is known exactly; m = min over the tile. In a later pass t the labels of the tile provably cannot change while
    m > sum_{t0 < tau <= t} 2 * max_j ||c_j^tau - c_j^(tau-1)||
(each centroid move shifts a distance by at most its length). Skipped tiles keep their labels and their sums.
This script runs the SPEC.md §4 schedule with the NumPy oracle and reports, per pass, the fraction of tiles the test
would skip (tile = 4 consecutive 8x8 blocks in raster-of-blocks order, as in the slab), on the BSD fixtures (per-image
codebooks) and on the synthetic bench batch (global codebook over 8 images)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from oracle import spec_oracle as so, c_oracle as co
from gabor_color_image_segmentation_amd.synthetic import synthetic_batch

def tile_ids(h, w):
    by, bx = np.mgrid[:h, :w]
    blk = (by // 8) * ((w + 7) // 8) + bx // 8
    return (blk // 4).ravel()

def study(xs, shapes, k=8, n_iter=10, name=""):
    x = np.concatenate(xs).astype(np.int64)
    tid = np.concatenate([tile_ids(h, w) + off for (h, w), off in zip(shapes, np.cumsum([0] + [((h + 7) // 8) * ((w + 7) // 8) // 4 + 1 for h, w in shapes[:-1]]))])
    ntile = tid.max() + 1
    c = so.kmeans_init(xs[0], k)
    m_tile = np.full(ntile, -1.0)          # min slack when last processed
    drift = np.zeros(ntile)                # accumulated 2*max move since then
    lab = None
    rows = []
    for t in range(n_iter):
        d2 = ((x[:, None, :] - c[None, :, :]) ** 2).sum(axis=2).astype(np.float64)
        new_lab = d2.argmin(axis=1)
        skippable = (m_tile > drift) if t > 0 else np.zeros(ntile, bool)
        if lab is not None:
            # sanity: labels of skippable tiles really do not change
            assert not np.any((new_lab != lab) & skippable[tid])
        part = np.partition(d2, 1, axis=1) if k > 1 else np.stack([d2[:, 0], np.full(len(d2), np.inf)], 1)
        slack = np.sqrt(part[:, 1]) - np.sqrt(part[:, 0])
        mt = np.full(ntile, np.inf)
        np.minimum.at(mt, tid, slack)
        processed = ~skippable
        m_tile[processed] = mt[processed]
        drift[processed] = 0.0
        changed = 0 if lab is None else float((new_lab != lab).mean())
        lab = new_lab
        rows.append((t, float(skippable.mean()), changed))
        if t < n_iter - 1:
            newc, _, _ = so.kmeans_update(x, lab, c)
            move = np.sqrt(((newc - c) ** 2).sum(axis=1).astype(np.float64))
            drift += 2.0 * move.max()
            c = newc
    print(name, " ".join(f"p{t}:{s:.0%}(chg {ch:.1%})" for t, s, ch in rows),
          f"| passes' traffic with skipping: {1 - sum(s for _, s, _ in rows) / n_iter:.2f}x")

tapq, shift = so.bank()
inp = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests", "golden", "bsd_inputs.npz"))
for i in inp["ids"]:
    img = inp["img_" + str(i)]
    f = co.gabor_features(img, tapq, shift, 6)
    study([f.reshape(72, -1).T], [img.shape[:2]], name=f"BSD {i} per-image")
imgs = synthetic_batch(8, 321, 481, seed=0)
fs = [co.gabor_features(im, tapq, shift, 6).reshape(72, -1).T for im in imgs]
study(fs[:1], [(321, 481)], name="synthetic image 0 per-image")
study(fs, [(321, 481)] * 8, name="synthetic 8 images global")
