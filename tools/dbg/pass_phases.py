#!/usr/bin/env python3
"""Where does a wave of the narrow Lloyd pass spend a tile? (round 6)

    bash tools/build_variant.sh phases -DGCS_KP_PHASES && python tools/dbg/pass_phases.py build_ab/phases.so

Runs the bench configuration's Lloyd pass (batch 64, 481 x 321, 4x6 bank, global codebook) on a library built with
-DGCS_KP_PHASES and prints, per phase, the shader-clock cycles per tile averaged over all waves (and the slowest / fastest
workgroup's): staging writes | first barrier | next tile's loads | assign | update + advance | second barrier."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gabor_color_image_segmentation_amd import _lib   # noqa: E402

_lib.LIB_PATH = os.path.abspath(sys.argv[1])
from gabor_color_image_segmentation_amd import Segmenter                            # noqa: E402
from gabor_color_image_segmentation_amd.synthetic import synthetic_shard            # noqa: E402

B, H, W = 64, 321, 481
imgs = torch.from_numpy(synthetic_shard(0, B, H, W)).cuda()
seg = Segmenter()
ws = seg._workspace(B, H, W, "global")
seg.ops.gabor_features(imgs, ws["feats"])
seg.ops.kmeans_init(ws["feats"], B, H, W, seg.k, 1, ws["cent"])
raw = ctypes.CDLL(_lib.LIB_PATH)
out = np.zeros(1024 * 4 * 8, np.uint64)
names = ["stage_write", "barrier 1", "loads issue", "assign", "update+advance", "barrier 2"]
for rev in (False, True, False, True):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    seg.ops.assign_accumulate(ws["feats"], ws["cent"], B, H, W, seg.k, 1, None, ws["partials"], reverse=rev)
    e1.record()
    torch.cuda.synchronize()
    assert raw.gcs_debug_kp_phases(ctypes.c_void_p(out.ctypes.data)) == 0
    ph = out.reshape(1024, 4, 8)[:768].astype(np.float64)
    tiles = ph[:, :, 6]
    per_tile = ph[:, :, :6] / np.maximum(tiles[:, :, None], 1)
    tot = per_tile.sum(axis=2)
    print(f"pass ({'reverse' if rev else 'forward'}) {e0.elapsed_time(e1) * 1e3:.1f} us; tiles per workgroup {tiles.mean():.1f}; "
          f"cycles per tile and wave: mean {tot.mean():.0f} (workgroup min {tot.mean(axis=1).min():.0f}, max {tot.mean(axis=1).max():.0f})")
    for k, n in enumerate(names):
        v = per_tile[:, :, k]
        print(f"   {n:16s} mean {v.mean():7.0f}   per wave 0..3: " + " ".join(f"{v[:, w].mean():7.0f}" for w in range(4)))
