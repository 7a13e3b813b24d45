#!/usr/bin/env python3
"""Times the small kernels of a step (reduce+finalize, pad via the Gabor entry is not separable, widen, init)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gabor_color_image_segmentation_amd import Segmenter
from gabor_color_image_segmentation_amd.synthetic import synthetic_shard
B, H, W = 64, 321, 481
imgs = torch.from_numpy(synthetic_shard(0, B, H, W)).cuda()
seg = Segmenter()
ws = seg._workspace(B, H, W, "global")
seg.ops.gabor_features(imgs, ws["feats"])
seg.ops.kmeans_init(ws["feats"], B, H, W, 8, 1, ws["cent"])
seg.ops.assign_accumulate(ws["feats"], ws["cent"], B, H, W, 8, 1, ws["labels"], ws["partials"])
out = torch.empty((B, H, W), dtype=torch.int32, device="cuda")


def timed(name, fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    print(f"{name:28s} {s.elapsed_time(e) / n * 1e3:7.1f} us per call (back to back)")


timed("reduce_finalize", lambda: seg.ops.reduce_finalize(ws["partials"], B, H, W, 8, 1, ws["sums"], ws["cent"]))
timed("reduce", lambda: seg.ops.reduce(ws["partials"], B, H, W, 8, 1, ws["sums"]))
timed("finalize", lambda: seg.ops.finalize(ws["sums"], 1, 8, ws["cent"]))
timed("labels_widen", lambda: seg.ops.labels_widen(ws["labels"], B, H, W, out))
timed("kmeans_init", lambda: seg.ops.kmeans_init(ws["feats"], B, H, W, 8, 1, ws["cent"]))
