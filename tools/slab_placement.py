#!/usr/bin/env python3
"""Does the Lloyd pass time depend on WHICH 0.9 GB allocation holds the feature slab? (profiles/r2_notes.md: builds in one
tools/ab.py process differ by 4-7 % with identical code.) Allocates N slabs, fills each with the same features (device
copy of slab 0), and times alternating-direction passes on each, interleaved over rounds. Prints address and median."""
import os, sys, statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gabor_color_image_segmentation_amd import Segmenter
from gabor_color_image_segmentation_amd.synthetic import synthetic_shard
N = int(sys.argv[1]) if len(sys.argv) > 1 else 6
B, H, W = 64, 321, 481
imgs = torch.from_numpy(synthetic_shard(0, B, H, W)).cuda()
seg = Segmenter()
ws = seg._workspace(B, H, W, "global")
seg.ops.gabor_features(imgs, ws["feats"])
seg.ops.kmeans_init(ws["feats"], B, H, W, seg.k, 1, ws["cent"])
slabs = [ws["feats"]]
hold = []
for i in range(1, N):
    if i % 2 == 0:
        hold.append(torch.empty(300 << 20, dtype=torch.uint8, device="cuda"))     # perturb the allocator between slabs
    s = torch.empty_like(ws["feats"]); s.copy_(ws["feats"]); slabs.append(s)
times = [[] for _ in slabs]
for rnd in range(10):
    for i, s in enumerate(slabs):
        for rev in (False, True, False, True):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            seg.ops.assign_accumulate(s, ws["cent"], B, H, W, seg.k, 1, ws["labels"], ws["partials"], reverse=rev)
            e1.record(); torch.cuda.synchronize()
            if rnd >= 2 and rev:
                times[i].append(e0.elapsed_time(e1))
for i, s in enumerate(slabs):
    a = s.data_ptr()
    print(f"slab {i}: addr 0x{a:x} (mod 1GiB 0x{a % (1 << 30):x}, mod 2MiB 0x{a % (2 << 20):x})  warm pass median {statistics.median(times[i]):.4f} ms  min {min(times[i]):.4f}")
