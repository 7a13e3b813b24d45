#!/usr/bin/env python3
"""Does the Gabor stage of the NEXT batch overlap the Lloyd passes of the current one when they run on two streams?
Times, per 64-image batch: the sequential step, and a two-stream software pipeline over N batches (two feature slabs:
stream A runs gabor(n+1) while stream B runs the ten passes of batch n). Same kernels, same results."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gabor_color_image_segmentation_amd import Segmenter
from gabor_color_image_segmentation_amd.segmenter import lloyd
from gabor_color_image_segmentation_amd.synthetic import synthetic_shard

B, H, W = 64, 321, 481
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
imgs = torch.from_numpy(synthetic_shard(0, B, H, W)).cuda()
seg = Segmenter()
ops = seg.ops
out = torch.empty((B, H, W), dtype=torch.int32, device="cuda")
for _ in range(15):
    seg.segment_device(imgs, mode="global", out=out)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(N):
    seg.segment_device(imgs, mode="global", out=out)
torch.cuda.synchronize()
seq = (time.perf_counter() - t0) / N * 1e3
want = out.clone()

ws = [seg._tail_workspace(B, H, W, "global") for _ in range(2)]
outs = [torch.empty_like(out) for _ in range(2)]
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
ev_feat = [torch.cuda.Event() for _ in range(2)]
ev_free = [torch.cuda.Event() for _ in range(2)]


def pipeline(n):
    for i in range(n + 1):
        s = i & 1
        if i < n:
            with torch.cuda.stream(sa):
                if i >= 2:
                    sa.wait_event(ev_free[s])          # the passes of batch i-2 have finished with slab s
                ops.gabor_features(imgs, ws[s]["feats"])
                ev_feat[s].record(sa)
        if i >= 1:
            p = (i - 1) & 1
            with torch.cuda.stream(sb):
                sb.wait_event(ev_feat[p])
                lloyd(ops, ws[p]["feats"], B, H, W, seg.k, seg.n_iter, "global", ws[p]["labels"], ws[p]["partials"],
                      ws[p]["cent"], ws[p]["sums"], raster=outs[p])
                ev_free[p].record(sb)


pipeline(6)
torch.cuda.synchronize()
t0 = time.perf_counter()
pipeline(N)
torch.cuda.synchronize()
pip = (time.perf_counter() - t0) / N * 1e3
print(f"sequential step {seq:.3f} ms ({B*H*W/seq/1e3:.0f} Mpix/s)   two-stream pipeline {pip:.3f} ms per batch ({B*H*W/pip/1e3:.0f} Mpix/s)")
print("results equal:", bool(torch.equal(outs[0], want)) and bool(torch.equal(outs[1], want)))
