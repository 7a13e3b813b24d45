"""Step 4 (run under /opt/conda/bin/python3.9, the interpreter that has scikit-image).

The reference tree has no Gabor code, but it is a scikit-image program (script.py:9-11 import
skimage) and scikit-image publishes a Gabor kernel. This script stores
``skimage.filters.gabor_kernel(frequency, theta, bandwidth)`` for every (scale, orientation) of the
default bank and of one 8x8 bank (at the base frequency f_s * 2^(s//2) of the filter's pyramid level), so that SPEC.md §2's envelope, bandwidth -> sigma rule, rotation
convention and phase are pinned to a published definition (tests/test_bank.py). Kernels are stored
on a fixed 31x31 frame centred on the origin (zero outside skimage's own support).
"""
import math
import os
import sys

import numpy as np

np.complex = complex      # skimage 0.18 still spells the dtype with the alias numpy 1.24 removed
from skimage.filters import gabor_kernel   # noqa: E402

R = 15
out = {}
for name, (ns, no) in {"b4x6": (4, 6), "b8x8": (8, 8)}.items():
    frames = np.zeros((ns * no, 2 * R + 1, 2 * R + 1), np.complex128)
    for s in range(ns):
        freq = 0.4 / math.sqrt(2.0) ** s * 2.0 ** (s // 2)       # f_base on pyramid level s // 2 (SPEC.md §2)
        for o in range(no):
            g = gabor_kernel(freq, theta=o * math.pi / no, bandwidth=1.0)
            ry, rx = g.shape[0] // 2, g.shape[1] // 2
            cy, cx = min(ry, R), min(rx, R)
            frames[s * no + o, R - cy:R + cy + 1, R - cx:R + cx + 1] = g[ry - cy:ry + cy + 1, rx - cx:rx + cx + 1]
    out[name] = frames
    out[name + "_shape"] = np.array([ns, no])
import skimage  # noqa: E402
out["skimage_version"] = np.array(skimage.__version__)
np.savez_compressed(os.path.join(sys.argv[1], "bank_skimage.npz"), **out)
print("bank_skimage.npz", {k: v.shape for k, v in out.items()})
