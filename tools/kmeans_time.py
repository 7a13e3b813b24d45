#!/usr/bin/env python3
"""Time one Lloyd pass (gcs_kmeans_assign_accumulate) on 64 x 321x481 features for GCS_LIB_PATH."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gabor_color_image_segmentation_amd import Segmenter
from gabor_color_image_segmentation_amd.synthetic import synthetic_shard
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
mode = sys.argv[2] if len(sys.argv) > 2 else "global"
imgs = torch.from_numpy(synthetic_shard(0, B, 321, 481)).cuda()
seg = Segmenter()
ws = seg._workspace(B, 321, 481, mode)
n_sets = B if mode == "per_image" else 1
seg.ops.gabor_features(imgs, ws["feats"])
seg.ops.kmeans_init(ws["feats"], B, 321, 481, 8, n_sets, ws["cent"])
for _ in range(3):
    seg.ops.assign_accumulate(ws["feats"], ws["cent"], B, 321, 481, 8, n_sets, ws["labels"], ws["partials"])
torch.cuda.synchronize()
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
for s, e in ev:
    s.record(); seg.ops.assign_accumulate(ws["feats"], ws["cent"], B, 321, 481, 8, n_sets, ws["labels"], ws["partials"]); e.record()
torch.cuda.synchronize()
t = sorted(s.elapsed_time(e) for s, e in ev)
gb = (2 * 72 + 1) * B * 321 * 481 / 1e9
print(f"{os.environ.get('GCS_LIB_PATH','default')[-24:]:24s} pass B={B} {mode}: median {t[10]*1e3:.1f} us  min {t[0]*1e3:.1f} us -> {gb/t[10]*1e3:.0f} GB/s (alg)")
