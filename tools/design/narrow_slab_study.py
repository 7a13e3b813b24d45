#!/usr/bin/env python3
"""Design study: how many feature values need more than 12 bits, and how are they clustered?

The Lloyd passes stream `uint16` features (SPEC.md §3: g <= 46 163) although almost every value is below 4096. The candidate
format splits a value into a 12-bit BASE (low byte + a nibble, always read) and a 4-bit TOP nibble (bits 12..15) that a pass
reads only for tiles flagged "some top nibble is non-zero". Exactness then does not depend on the data: a flagged tile is
read at full width. What the format buys is decided by the fraction of FLAGGED TILES (a tile = four consecutive 8x8 blocks =
the 256 pixels of one k-means step, csrc/common.h), which this script counts with the C oracle for

  * the BSD500 `val` fixtures (tests/golden/bsd_val_images.npz: 24 decoded images, both orientations),
  * the synthetic bench batch (synthetic_batch(64): the images bench.py times), 4x6 and 8x8 banks,
  * a full-contrast square grating at f = 0.4 (the worst case the VERDICT names).

Per-pass bytes per pixel: wide = 2 * sum_L D_L / 4^L; narrow = 1.5 * that + 0.5 * that * flagged-tile fraction.
Run: python tools/design/narrow_slab_study.py [--bench-images N]   (CPU only; ~1 min)"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from oracle import spec_oracle as so, c_oracle as co                       # noqa: E402
from gabor_color_image_segmentation_amd.synthetic import synthetic_batch    # noqa: E402


def tile_flags(feat, thr):
    """feat (D, H, W) uint16 (levels replicated to full resolution) -> (flagged tiles, tiles, values >= thr, values)."""
    d, h, w = feat.shape
    hp, wp = (h + 7) // 8 * 8, (w + 7) // 8 * 8
    big = np.zeros((hp, wp), bool)
    big[:h, :w] = (feat >= thr).any(axis=0)
    blk = big.reshape(hp // 8, 8, wp // 8, 8).any(axis=(1, 3)).ravel()       # per 8x8 block, raster order of blocks
    nt = (blk.size + 3) // 4
    blk = np.concatenate([blk, np.zeros(4 * nt - blk.size, bool)])
    return int(blk.reshape(nt, 4).any(axis=1).sum()), nt, int((feat >= thr).sum()), feat.size


def bytes_per_px(n_scales, n_orient):
    per = 0.0
    for L in range((n_scales + 1) // 2):
        scales = 2 if n_scales - 2 * L >= 2 else 1
        per += 3 * scales * n_orient / 4 ** L
    return 2 * per


def report(name, imgs, n_scales, n_orient):
    tapq, shift = so.bank(n_scales=n_scales, n_orient=n_orient)
    ft = nt = nv = tot = 0
    ft13 = nv13 = 0
    vmax = 0
    for img in imgs:
        f = co.gabor_features(img, tapq, shift, n_orient)
        a, b, c, d = tile_flags(f, 4096)
        ft, nt, nv, tot = ft + a, nt + b, nv + c, tot + d
        a, _, c, _ = tile_flags(f, 8192)
        ft13, nv13 = ft13 + a, nv13 + c
        vmax = max(vmax, int(f.max()))
    wide = bytes_per_px(n_scales, n_orient)
    frac = ft / nt
    narrow = 0.75 * wide + 0.25 * wide * frac
    print(f"{name:34s} {n_scales}x{n_orient}  values>=4096 {nv:9d} of {tot:11d} ({nv / tot:.2e})  max {vmax:5d}  "
          f"tiles flagged {ft:6d} of {nt:6d} ({frac:6.2%})  [>=8192: values {nv13}, tiles {ft13}]  "
          f"pass B/px {wide:.1f} -> {narrow:.1f} ({narrow / wide:.3f})")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bench-images", type=int, default=16)
    a = ap.parse_args()
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
    z = np.load(os.path.join(root, "tests", "golden", "bsd_val_images.npz"))
    val = [z["img_" + str(i)] for i in z["ids"]]
    report("BSD val fixtures (24 images)", val, 4, 6)
    report("BSD val fixtures (8 images)", val[:8], 8, 8)
    bench = synthetic_batch(64)[:a.bench_images]
    report(f"bench batch (first {a.bench_images} of 64)", bench, 4, 6)
    report("bench batch (first 4 of 64)", bench[:4], 8, 8)
    yy, xx = np.mgrid[0:321, 0:481]
    for f0, nm in ((0.4, "square grating f=0.4, contrast 255"), (0.2828, "square grating f=0.283")):
        g = (np.sin(2 * np.pi * f0 * xx) >= 0).astype(np.uint8) * 255
        report(nm, [np.stack([g, g, g], -1)], 4, 6)
    rng = np.random.default_rng(1)
    report("uniform noise 0..255", [rng.integers(0, 256, (321, 481, 3), dtype=np.uint8)], 4, 6)


if __name__ == "__main__":
    main()
