import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gabor_color_image_segmentation_amd import Segmenter
from gabor_color_image_segmentation_amd.synthetic import synthetic_shard
for (B,H,W) in [(768,16,16),(64,321,481)]:
    imgs = torch.from_numpy(synthetic_shard(0, B, H, W)).cuda()
    seg = Segmenter()
    ws = seg._workspace(B, H, W, "global")
    seg.ops.gabor_features(imgs, ws["feats"])
    seg.ops.kmeans_init(ws["feats"], B, H, W, 8, 1, ws["cent"])
    f = lambda: seg.ops.assign_accumulate(ws["feats"], ws["cent"], B, H, W, 8, 1, ws["labels"], ws["partials"])
    for _ in range(3): f()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
    for s, e in ev:
        s.record(); f(); e.record()
    torch.cuda.synchronize()
    t = sorted(s.elapsed_time(e) for s, e in ev)
    print(f"B={B} {H}x{W}: parts={seg.ops.lib.gcs_kmeans_parts_per_image(B,H,W)} pass median {t[10]*1e3:.1f} us min {t[0]*1e3:.1f} us")
