import gc, os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gabor_color_image_segmentation_amd import Segmenter
from gabor_color_image_segmentation_amd.synthetic import synthetic_batch
def say(*a): print(*a, flush=True)
for trial, shapes in enumerate([[(1, 96, 128)], [(1, 321, 481)], [(1, 321, 481), (1, 481, 321), (3, 64, 88)], [(6, 72, 104)]]):
    seg = Segmenter(n_iter=3)
    for sh in shapes:
        imgs = synthetic_batch(*sh, seed=5)
        for mode in ("per_image", "global"):
            a = seg.segment_batch(imgs, mode=mode); b = seg.segment_batch(imgs, mode=mode)
            assert np.array_equal(a, b)
    torch.cuda.synchronize()
    say("trial", trial, shapes, "graphs", len(seg._graphs))
    del seg
    say(" collecting"); n = gc.collect(); say(" collected", n)
    torch.cuda.synchronize(); say(" sync ok")
say("done")
