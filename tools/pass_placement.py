#!/usr/bin/env python3
"""Does the pass time depend on where the feature slab sits? One process, one big buffer, the slab placed
at several byte offsets inside it (and in fresh allocations); 20 passes timed per placement."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gabor_color_image_segmentation_amd import Segmenter
from gabor_color_image_segmentation_amd.synthetic import synthetic_shard
B, H, W = 64, 321, 481
imgs = torch.from_numpy(synthetic_shard(0, B, H, W)).cuda()
seg = Segmenter()
ws = seg._workspace(B, H, W, "global")
nbytes = ws["feats"].numel() * ws["feats"].element_size()
print("slab bytes", nbytes, "dtype", ws["feats"].dtype, "base %x" % ws["feats"].data_ptr())


def time_pass(feats):
    seg.ops.gabor_features(imgs, feats)
    seg.ops.kmeans_init(feats, B, H, W, 8, 1, ws["cent"])
    f = lambda r: seg.ops.assign_accumulate(feats, ws["cent"], B, H, W, 8, 1, ws["labels"], ws["partials"], reverse=r)
    for i in range(4): f(i & 1)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
    for i, (s, e) in enumerate(ev):
        s.record(); f(bool(i & 1)); e.record()
    torch.cuda.synchronize()
    t = sorted(s.elapsed_time(e) for s, e in ev)
    return t[10] * 1e3, t[0] * 1e3


print("workspace slab: median %.1f us min %.1f us" % time_pass(ws["feats"]))
big = torch.empty(nbytes + (256 << 20), dtype=torch.uint8, device="cuda")
for off in [0, 4096, 65536, 1 << 20, (2 << 20) + 4096, 16 << 20, (33 << 20) + 8192, 100 << 20, 255 << 20]:
    v = big[off:off + nbytes].view(ws["feats"].dtype)
    print("offset %10d (ptr %x): median %.1f us min %.1f us" % ((off, v.data_ptr()) + time_pass(v)))
for i in range(4):
    fresh = torch.empty(nbytes, dtype=torch.uint8, device="cuda").view(ws["feats"].dtype)
    print("fresh allocation %d (ptr %x): median %.1f us min %.1f us" % ((i, fresh.data_ptr()) + time_pass(fresh)))
    keep = fresh  # noqa
