#!/usr/bin/env python3
"""Design study: which K-steps of the Gabor A operand multiply zeros?

gcs_bank_pack splits every Q15 tap into two signed byte digits (SPEC.md §2: lo = ((q + 128) & 255) - 128, hi = (q - lo) >> 8).
The envelope makes the taps small away from the kernel centre, so the HI digit is zero outside a central window. A 32-row A tile
holds the four digits {re_lo, re_hi, im_lo, im_hi} of four filters; K-step kk of gabor_mfma_kernel covers frame rows 2 kk, 2 kk + 1
(csrc/abi.hip: the 13 x 13 kernel sits at frame rows / columns 1 .. 13 of the 15 x 16 frame; KS = 7 K-steps cover frame rows 1 .. 14
as (1, 2), (3, 4), ... with the kernel's h = 0 / 1 halves). This script prints, per bank, the window of non-zero hi digits and the
MFMAs per (pixel pair, 12 filters) for three row groupings:
  today           3 mixed tiles x 7 K-steps                                   = 21
  hi rows apart   [lo f0-7] 7 + [lo f8-11 | hi f8-11] 7 + [hi f0-7] K-steps with a non-zero hi row only
  hi tile 9x9     the same with the hi tile re-bound to the rows of its window
What it costs is in profiles/r6_notes.md: the digits of one output then sit in TWO accumulator tuples."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from oracle import spec_oracle as so   # noqa: E402


def study(n_scales, n_orient, ksize=13):
    tapq, shift = so.bank(n_scales=n_scales, n_orient=n_orient, ksize=ksize)      # (F, 2, ks, ks) int16
    q = tapq.astype(np.int64)
    lo = ((q + 128) & 255) - 128
    hi = (q - lo) >> 8
    F = q.shape[0]
    print(f"bank {n_scales}x{n_orient}, ksize {ksize}: {F} filters; |hi| <= {int(np.abs(hi).max())}; "
          f"non-zero hi taps {100.0 * float((hi != 0).mean()):.1f} %")
    for s in range(min(2, n_scales)):                                             # the two base scales (every level uses these)
        h = hi[s * n_orient:(s + 1) * n_orient]
        rows = np.nonzero((h != 0).any(axis=(0, 1, 3)))[0]
        cols = np.nonzero((h != 0).any(axis=(0, 1, 2)))[0]
        print(f"  base scale {s}: hi != 0 only in kernel rows {rows.min()}..{rows.max()}, columns {cols.min()}..{cols.max()}")
    # K-steps (frame row pairs) that hold a non-zero hi tap: kernel row r sits at frame row r + 1; K-step kk = frame rows 2 kk + 1, 2 kk + 2
    # for KS = 7 (rows 1 .. 14), i.e. kernel rows 2 kk, 2 kk + 1
    hr = np.nonzero((hi != 0).any(axis=(0, 1, 3)))[0]
    ksteps = sorted({int(r) // 2 for r in hr})
    print(f"  K-steps with a non-zero hi row: {ksteps} of 0..6 -> hi-only tile needs {len(ksteps)} MFMAs instead of 7")
    today = 3 * 7
    apart = 7 + 7 + len(ksteps)
    win = int(hr.max() - hr.min() + 1)
    rebound = 7 + 7 + (win + 1) // 2
    print(f"  MFMAs per pixel pair and level: today {today}; hi rows apart {apart}; hi tile re-bound to its {win} rows {rebound}")


for ns, no in ((4, 6), (8, 8)):
    study(ns, no)
