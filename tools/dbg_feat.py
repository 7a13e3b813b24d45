import numpy as np, torch, sys
sys.path.insert(0,'/root/repo')
from gabor_color_image_segmentation_amd import Segmenter
from gabor_color_image_segmentation_amd.synthetic import synthetic_batch
from oracle import spec_oracle as so
for (ns,no,ks) in [(2,3,7),(4,6,13)]:
    imgs = synthetic_batch(1, 50, 70, seed=5)
    seg = Segmenter(n_scales=ns, n_orient=no, ksize=ks)
    got = seg.features_device(torch.from_numpy(imgs).cuda()).cpu().numpy().view(np.uint16)[0]
    tapq, shift = so.bank(ns, no, ks)
    ref = so.gabor_features(imgs[0], tapq, shift, no)
    bad = np.argwhere(got != ref)
    print((ns,no,ks), "bad", len(bad), "of", got.size)
    if len(bad):
        d,y,x = bad[:,0],bad[:,1],bad[:,2]
        print(" planes", np.unique(d)[:40], " x mod 8 hist", np.bincount(x%8, minlength=8), " y mod 8", np.bincount(y%8,minlength=8))
        for b_ in bad[:6]: print("  ", b_, got[tuple(b_)], ref[tuple(b_)])
