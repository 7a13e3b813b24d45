#!/usr/bin/env python3
"""Lloyd pass time by sweep direction history (batch 64): what the Infinity Cache contributes.
After a Gabor stage: reverse, forward, forward, forward, reverse, reverse - a pass in the SAME direction as the one before
finds nothing of the 0.9 GB slab in the 256 MiB cache (pure HBM), one in the opposite direction starts on what was read last."""
import os, sys, statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gabor_color_image_segmentation_amd import Segmenter
from gabor_color_image_segmentation_amd.synthetic import synthetic_shard
B, H, W = 64, 321, 481
imgs = torch.from_numpy(synthetic_shard(0, B, H, W)).cuda()
seg = Segmenter()
ops = seg.ops
ws = seg._workspace(B, H, W, "global")
seq = [True, False, False, False, True, True, False, True]
acc = [[] for _ in seq]
for it in range(14):
    ops.gabor_features(imgs, ws["feats"])
    ops.kmeans_init(ws["feats"], B, H, W, seg.k, 1, ws["cent"])
    evs = []
    for rev in seq:
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        ops.assign_accumulate(ws["feats"], ws["cent"], B, H, W, seg.k, 1, None, ws["partials"], reverse=rev)
        e.record()
        evs.append((s, e))
    torch.cuda.synchronize()
    if it >= 4:
        for i, (s, e) in enumerate(evs):
            acc[i].append(s.elapsed_time(e))
print("after gabor: " + "  ".join(("rev" if r else "fwd") + f" {statistics.median(a):.4f}" for r, a in zip(seq, acc)))
