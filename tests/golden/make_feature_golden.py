"""Step 5 (run under /opt/conda/bin/python3.9, the interpreter that has scikit-image).

scikit-image's own Gabor FILTER (skimage.filters.gabor: ndi.convolve of the image with the real and imaginary
gabor_kernel, mode='reflect') on a crop of the first fixture image, red channel, for the two finest scales of the
default bank (their 3-sigma support fits the 15x15 frame of SPEC.md). Stores float32 magnitudes; tests/test_oracle.py
compares the oracle's Q7 features with them (same filter up to the unit-DC gain, the 15x15 truncation and the
fixed-point rounding).
"""
import math
import os
import sys

import numpy as np

np.complex = complex      # skimage 0.18 still spells the dtype with the alias numpy 1.24 removed
from skimage.filters import gabor   # noqa: E402

here = sys.argv[1]
inp = np.load(os.path.join(here, "bsd_inputs.npz"))
i = str(inp["ids"][0])
y0, x0, h, w = 40, 60, 96, 128
crop = inp["img_" + i][y0:y0 + h, x0:x0 + w]
chan = crop[:, :, 0].astype(np.float64)
mags = np.zeros((12, h, w), np.float32)
for s in range(2):
    freq = 0.4 / math.sqrt(2.0) ** s
    for o in range(6):
        re, im = gabor(chan, frequency=freq, theta=o * math.pi / 6, bandwidth=1.0, mode="reflect")
        mags[s * 6 + o] = np.sqrt(re * re + im * im)
np.savez_compressed(os.path.join(here, "features_skimage.npz"), id=np.array(i), box=np.array([y0, x0, h, w]),
                    crop=crop, magnitude=mags)
print("features_skimage.npz", mags.shape, float(mags.max()))
