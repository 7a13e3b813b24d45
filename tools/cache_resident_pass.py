#!/usr/bin/env python3
"""VERDICT r3 item 3 / SURVEY §7.3-5: is a feature slab that fits the 256 MiB Infinity Cache streamed faster by the Lloyd
passes that re-read it? For groups of g BSD-sized images (per-image codebooks, the reference's semantics): slab size, the
time of a pass that follows another pass on the same slab (what a resident slab would speed up), microseconds per MB beside
the 64-image figure, and the whole per-image step of 64 images cut into groups of g (`segment_device(group=g)`).

    python tools/cache_resident_pass.py [g ...]        (default 2 4 8 12 16 24 32 64)
"""
import os, statistics, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gabor_color_image_segmentation_amd import Segmenter
from gabor_color_image_segmentation_amd.synthetic import synthetic_shard

H, W, K = 321, 481, 8
groups = [int(x) for x in sys.argv[1:]] or [2, 4, 8, 12, 16, 24, 32, 64]
seg = Segmenter()
imgs64 = torch.from_numpy(synthetic_shard(0, 64, H, W)).cuda()
out = torch.empty((64, H, W), dtype=torch.int32, device="cuda")


def ev():
    return torch.cuda.Event(enable_timing=True)


print(f"{'g':>3s} {'slab MB':>8s} {'pass us (same dir)':>19s} {'pass us (alternating)':>22s} {'us/MB':>7s} {'GB/s':>7s} | step of 64 images in groups of g: ms, Mpix/s")
for g in groups:
    imgs = imgs64[:g].contiguous()
    ws = seg._tail_workspace(g, H, W, "per_image")
    mb = ws["feats"].numel() / 1e6
    seg.ops.gabor_features(imgs, ws["feats"])
    seg.ops.kmeans_init(ws["feats"], g, H, W, K, g, ws["cent"])
    res = {}
    for name, alt in (("same", False), ("alt", True)):
        ts = []
        for rep in range(6):
            e = [ev() for _ in range(11)]
            for t in range(10):                                   # ten passes back to back on the same slab, sums only
                e[t].record()
                seg.ops.assign_accumulate(ws["feats"], ws["cent"], g, H, W, K, g, None, ws["partials"], reverse=bool(t & 1) if alt else False)
            e[10].record()
            torch.cuda.synchronize()
            if rep:
                ts += [e[t].elapsed_time(e[t + 1]) * 1e3 for t in range(2, 10)]   # passes that follow a pass
        res[name] = statistics.median(ts)
    best = min(res.values())
    for _ in range(3):
        seg.segment_device(imgs64, mode="per_image", out=out, group=g)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 6
    for _ in range(n):
        seg.segment_device(imgs64, mode="per_image", out=out, group=g)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    print(f"{g:3d} {mb:8.1f} {res['same']:19.1f} {res['alt']:22.1f} {best / mb:7.3f} {mb / best * 1e3:7.0f} | {ms:7.3f} ms {64 * H * W / ms / 1e3:7.0f} Mpix/s")
