"""Seeded synthetic BSD-shaped images.

Not i.i.d. noise (k-means would have nothing to find): each image is a Voronoi
partition into 4-8 regions, every region a base colour plus an oriented sinusoid of
random frequency / angle, plus sigma=8 Gaussian noise, clipped to uint8.
"""
from __future__ import annotations

import numpy as np


def synthetic_image(rng: np.random.Generator, h=321, w=481) -> np.ndarray:
    n_reg = int(rng.integers(4, 9))
    sites = rng.uniform(0, 1, (n_reg, 2)) * np.array([h, w])
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    d2 = (yy[None] - sites[:, 0, None, None].astype(np.float32)) ** 2 + \
         (xx[None] - sites[:, 1, None, None].astype(np.float32)) ** 2
    region = np.argmin(d2, axis=0)
    base = rng.uniform(40, 215, (n_reg, 3)).astype(np.float32)
    freq = rng.uniform(0.05, 0.4, n_reg).astype(np.float32)
    ang = rng.uniform(0, np.pi, n_reg).astype(np.float32)
    amp = rng.uniform(10, 40, (n_reg, 3)).astype(np.float32)
    ph = 2 * np.pi * freq[region] * (xx * np.cos(ang[region]) + yy * np.sin(ang[region]))
    img = base[region] + amp[region] * np.sin(ph)[..., None]
    img += rng.normal(0, 8, img.shape).astype(np.float32)
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def synthetic_batch(b: int, h=321, w=481, seed=0) -> np.ndarray:
    """(B,H,W,3) uint8, deterministic in (b, h, w, seed): image i depends only on
    (seed, i), so a rank can generate its own shard of a global batch."""
    out = np.empty((b, h, w, 3), np.uint8)
    for i in range(b):
        out[i] = synthetic_image(np.random.default_rng([seed, i]), h, w)
    return out


def synthetic_shard(first: int, count: int, h=321, w=481, seed=0) -> np.ndarray:
    """Images [first, first+count) of the global batch with this seed."""
    out = np.empty((count, h, w, 3), np.uint8)
    for i in range(count):
        out[i] = synthetic_image(np.random.default_rng([seed, first + i]), h, w)
    return out
