import os, sys, torch
sys.path.insert(0, "/root/repo")
from gabor_color_image_segmentation_amd import Segmenter
from gabor_color_image_segmentation_amd.synthetic import synthetic_shard
for ns, no in ((4, 6), (8, 8)):
    imgs = torch.from_numpy(synthetic_shard(0, 64, 321, 481)).cuda()
    seg = Segmenter(n_scales=ns, n_orient=no)
    def timeit(n=15):
        for _ in range(3): seg.segment_device(imgs, mode="global")
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): seg.segment_device(imgs, mode="global")
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n
    res = []
    for rnd in range(3):
        a = timeit()
        real = seg.ops.reduce_finalize
        seg.ops.reduce_finalize = lambda *a, **k: None
        b = timeit()
        seg.ops.reduce_finalize = real
        res.append((a, b))
    print(f"bank {ns}x{no}: step with / without the 9 reduce_finalize launches: " + "  ".join(f"{a:.3f} / {b:.3f} ms (-{(a-b)*1e3:.0f} us)" for a, b in res))
