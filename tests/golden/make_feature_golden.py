"""Step 5 (run under /opt/conda/bin/python3.9, the interpreter that has scikit-image).

Pins SPEC.md §3 to published code, filter by filter, for the default 4x6 bank and the 8x8 bank (BASELINE config 4):

* the pyramid: ``skimage.transform.downscale_local_mean(I_L, (2, 2))`` of the edge-padded level, rounded half up
  (stored as uint8 level images);
* every filter: scikit-image's own Gabor FILTER (``skimage.filters.gabor``: ``ndi.convolve`` of the image with the real
  and imaginary ``gabor_kernel``, mode='reflect') run on the pyramid level the filter belongs to, at the level's base
  frequency f_s * 2^L. Stored: float32 magnitudes sqrt(re^2 + im^2), red channel of a crop of the first fixture image
  (odd-sized, so the edge replication of the pyramid is exercised). Level 0 of the 8x8 bank is stored on a stride-2
  grid to keep the file small.

tests/test_oracle.py compares the oracle's Q7 features with them (same filter up to the unit-DC gain, skimage's
per-orientation support box and the fixed-point rounding).
"""
import math
import os
import sys

import numpy as np

np.complex = complex      # skimage 0.18 still spells the dtype with the alias numpy 1.24 removed
from skimage.filters import gabor   # noqa: E402
from skimage.transform import downscale_local_mean   # noqa: E402

here = sys.argv[1]
inp = np.load(os.path.join(here, "bsd_inputs.npz"))
i = str(inp["ids"][0])
img = inp["img_" + i]
out = dict(id=np.array(i))
for name, (ns, no, (y0, x0, h, w), stride0) in {"b4x6": (4, 6, (40, 60, 97, 131), 1),
                                                "b8x8": (8, 8, (30, 50, 129, 193), 2)}.items():
    crop = img[y0:y0 + h, x0:x0 + w]
    out[name + "_box"] = np.array([y0, x0, h, w])
    out[name + "_crop"] = crop
    out[name + "_stride0"] = np.array(stride0)
    level = crop[:, :, 0].astype(np.float64)
    for lv in range((ns + 1) // 2):
        if lv:
            padded = np.pad(level, ((0, level.shape[0] & 1), (0, level.shape[1] & 1)), mode="edge")
            level = np.floor(downscale_local_mean(padded, (2, 2)) + 0.5)
        out[f"{name}_level{lv}"] = level.astype(np.uint8)
        st = stride0 if lv == 0 else 1
        mags = []
        for s in range(2 * lv, min(ns, 2 * lv + 2)):
            f_base = 0.4 / math.sqrt(2.0) ** s * 2.0 ** lv
            for o in range(no):
                re, im = gabor(level, frequency=f_base, theta=o * math.pi / no, bandwidth=1.0, mode="reflect")
                mags.append(np.sqrt(re * re + im * im)[::st, ::st].astype(np.float32))
        out[f"{name}_mag{lv}"] = np.stack(mags)
        print(name, "level", lv, level.shape, out[f"{name}_mag{lv}"].shape, float(out[f"{name}_mag{lv}"].max()))
np.savez_compressed(os.path.join(here, "features_skimage.npz"), **out)
