"""Gabor filter bank in fixed point on an octave pyramid (SPEC.md §2).

The reference ships no Gabor code; this is the build-authored
bank that fills the segmenter slot at /root/reference/BSD_metrics/script.py:30.
Host-side only (numpy, float64 -> int16); the device receives the packed
int8 MFMA digits produced by ``gcs_bank_pack`` (include/gcs.h).
"""
from __future__ import annotations

import dataclasses
import math

import numpy as np

KSIZE_MAX = 15          # the HIP kernel's tap frame is 15 rows x 16 columns
N_SCALES_MAX = 8        # octave pyramid of at most 4 levels (scales 2L, 2L+1 run on level L)
TAPQ_MAX = 32639        # 127*256 + 127: largest value two signed byte digits hold
FEATURE_Q = 7           # features are Q7 grey levels
TAP_Q = 15              # taps are Q15 (less only for kernels so peaked that a tap would not fit two byte digits)


@dataclasses.dataclass(frozen=True)
class GaborBank:
    """Quantised bank. ``tapq`` is ``[F, 2, ksize, ksize]`` int16 (re, im)."""
    n_scales: int
    n_orient: int
    ksize: int
    f_max: float
    ratio: float
    bandwidth: float
    exponent: int            # E: tapq = rint(tap * 2**E)
    shift: int               # E - FEATURE_Q
    tapq: np.ndarray

    @property
    def n_filters(self) -> int:
        return self.n_scales * self.n_orient

    @property
    def n_features(self) -> int:
        return 3 * self.n_filters

    @property
    def n_levels(self) -> int:
        """Pyramid levels the bank spans (SPEC.md §2): scales 2L and 2L+1 run on level L."""
        return (self.n_scales + 1) // 2


def gabor_taps(n_scales=4, n_orient=6, ksize=13, f_max=0.4, ratio=math.sqrt(2.0),
               bandwidth=1.0) -> np.ndarray:
    """Float64 taps ``[F, 2, ksize, ksize]`` (SPEC.md §2, before quantisation)."""
    if ksize % 2 != 1 or not (1 <= ksize <= KSIZE_MAX):
        raise ValueError(f"ksize must be odd and <= {KSIZE_MAX}, got {ksize}")
    if n_scales < 1 or n_orient < 1:
        raise ValueError("n_scales and n_orient must be >= 1")
    if n_scales > N_SCALES_MAX:
        raise ValueError(f"n_scales must be <= {N_SCALES_MAX} (4 pyramid levels)")
    r = (ksize - 1) // 2
    dy, dx = np.mgrid[-r:r + 1, -r:r + 1].astype(np.float64)
    kappa = math.sqrt(math.log(2.0) / 2.0) / math.pi * \
        (2.0 ** bandwidth + 1.0) / (2.0 ** bandwidth - 1.0)
    taps = np.empty((n_scales * n_orient, 2, ksize, ksize), np.float64)
    for s in range(n_scales):
        freq = f_max / ratio ** s * 2.0 ** (s // 2)      # f_base: cycles per pixel of pyramid level s // 2
        sigma = kappa / freq
        env = np.exp(-(dx * dx + dy * dy) / (2.0 * sigma * sigma))
        env /= env.sum()
        for o in range(n_orient):
            theta = o * math.pi / n_orient
            phase = 2.0 * math.pi * freq * (dx * math.cos(theta) + dy * math.sin(theta))
            taps[s * n_orient + o, 0] = env * np.cos(phase)
            taps[s * n_orient + o, 1] = env * np.sin(phase)
    return taps


def make_bank(n_scales=4, n_orient=6, ksize=13, f_max=0.4, ratio=math.sqrt(2.0),
              bandwidth=1.0) -> GaborBank:
    """Quantise the bank to Q15 taps (SPEC.md §2): shift = 8, i.e. the response's Q7 value is bytes 1..2 of v."""
    taps = gabor_taps(n_scales, n_orient, ksize, f_max, ratio, bandwidth)
    exponent = min(TAP_Q, int(math.floor(math.log2(TAPQ_MAX / np.abs(taps).max()))))
    tapq = np.rint(taps * 2.0 ** exponent).astype(np.int64)
    if np.abs(tapq).max() > TAPQ_MAX:
        raise AssertionError("tap quantisation overflowed the two-digit range")
    if exponent < FEATURE_Q:
        raise ValueError("bank too peaked for Q7 features")
    if np.any(tapq[:, 1].sum(axis=(1, 2)) != 0):
        raise AssertionError("imaginary taps must sum to zero (odd symmetry)")
    return GaborBank(n_scales, n_orient, ksize, float(f_max), float(ratio),
                     float(bandwidth), exponent, exponent - FEATURE_Q,
                     tapq.astype(np.int16))


def split_digits(tapq: np.ndarray):
    """tapq = 256*hi + lo with both digits in [-128, 127] (SPEC.md §2)."""
    q = tapq.astype(np.int32)
    lo = ((q + 128) & 255) - 128
    hi = (q - lo) >> 8
    if hi.min() < -128 or hi.max() > 127:
        raise AssertionError("high digit out of int8 range")
    return lo.astype(np.int8), hi.astype(np.int8)
