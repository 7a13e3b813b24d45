#!/usr/bin/env python3
"""Randomised parity sweep of the Gabor stage + label map against the C oracle: random shapes (odd sizes, tiny, one tile row of
a single valid row, ...), batch sizes, banks (1-8 scales, odd orientation counts, ksize 1-15), k. Half of the cases carry one to
three full-contrast square-wave patches (random period, direction, size, place): values of 4096 and more, i.e. flagged tiles of the
split slab (round 6) among clean ones, also on packed edge strips. Prints one line per case and a summary; exit code 1 on any
mismatch. usage: fuzz_features.py [n_cases] [seed]"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from gabor_color_image_segmentation_amd import Segmenter
from gabor_color_image_segmentation_amd.synthetic import synthetic_batch
from oracle import c_oracle as co

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
t0 = time.time()
for case in range(n_cases):
    ns = int(rng.choice([1, 2, 3, 4, 4, 4, 5, 6, 8]))
    no = int(rng.choice([1, 2, 3, 4, 5, 6, 6, 7, 8, 9]))
    ks = int(rng.choice([1, 3, 5, 7, 9, 11, 13, 13, 13, 15]))
    if ns * no * 3 > 207 and rng.random() < 0.7:
        no = max(1, 69 // ns)
    h = int(rng.choice([8, 9, 10, 17, 31, 32, 33, 34, 36, 63, 64, 65, 66, 97, 129, 161, 200, 321]))
    w = int(rng.choice([8, 10, 11, 24, 40, 42, 63, 64, 65, 72, 96, 130, 200, 257, 481, 482]))
    b = int(rng.choice([1, 1, 2, 3, 5, 9]))
    k = int(rng.choice([1, 2, 5, 8, 8, 11, 16]))
    n_iter = int(rng.choice([1, 2, 3]))
    mode = str(rng.choice(["per_image", "global"]))
    imgs = synthetic_batch(b, h, w, seed=int(rng.integers(1 << 30)))
    if rng.random() < 0.15:
        imgs[:] = rng.choice([0, 255])                 # extreme constant pixels
    n_patch = int(rng.integers(1, 4)) if rng.random() < 0.5 else 0
    for _ in range(n_patch):                           # full-contrast square waves: values >= 4096 under them
        size = int(rng.integers(4, 25))
        ph, pw = min(size, h), min(size, w)
        y0, x0 = int(rng.integers(0, h - ph + 1)), int(rng.integers(0, w - pw + 1))
        if rng.random() < 0.4:                         # push the patch onto the right / bottom edge (packed strips)
            y0, x0 = (h - ph, x0) if rng.random() < 0.5 else (y0, w - pw)
        yy, xx = np.mgrid[0:ph, 0:pw]
        ang = rng.random() * np.pi
        wave = np.sin(2 * np.pi * float(rng.choice([0.2, 0.3, 0.4, 0.5])) * (xx * np.cos(ang) + yy * np.sin(ang)) + 0.3) >= 0
        imgs[int(rng.integers(b)), y0:y0 + ph, x0:x0 + pw] = np.where(wave, 255, 0).astype(np.uint8)[..., None]
    try:
        seg = Segmenter(n_scales=ns, n_orient=no, ksize=ks, k=k, n_iter=n_iter)
    except Exception as e:                             # e.g. a degenerate bank the packer refuses
        print(f"case {case}: bank {ns}x{no} ks {ks}: {type(e).__name__}: {e}")
        continue
    bank = seg.bank
    feats = seg.features_device(torch.from_numpy(imgs).cuda()).cpu().numpy().view(np.uint16)
    ref = np.stack([co.gabor_features(im, bank.tapq, bank.shift, no) for im in imgs])
    ok_f = np.array_equal(feats, ref)
    lab = seg.segment_batch(imgs, mode=mode)
    want = co.segment_batch(imgs, bank.tapq, bank.shift, no, k=k, n_iter=n_iter, mode=mode)
    ok_l = np.array_equal(lab, want)
    big = int((ref >= 4096).sum())
    print(f"case {case}: B {b} {h}x{w} bank {ns}x{no} ks {ks} k {k} it {n_iter} {mode} patches {n_patch} values>=4096 {big}: features {'ok' if ok_f else 'MISMATCH'} labels {'ok' if ok_l else 'MISMATCH'}", flush=True)
    if not (ok_f and ok_l):
        bad += 1
        if not ok_f:
            q = np.argwhere(feats != ref)
            print("   first bad", q[:3].tolist(), "count", len(q), "planes", np.unique(q[:, 1])[:12].tolist())
print(f"{n_cases} cases, {bad} bad, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
