import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gabor_color_image_segmentation_amd import Segmenter
from gabor_color_image_segmentation_amd.synthetic import synthetic_batch
from oracle import c_oracle as co
seg = Segmenter(n_iter=3)
bank = seg.bank
for shape in [(3, 48, 80), (6, 72, 104), (1, 96, 128), (6, 200, 300), (2, 321, 481)]:
    imgs = synthetic_batch(*shape, seed=30)
    d = torch.from_numpy(imgs).cuda()
    print("shape", shape, flush=True)
    f = seg.features_device(d); torch.cuda.synchronize(); print(" features ok", flush=True)
    ref = co.gabor_features(imgs[0], bank.tapq, bank.shift, bank.n_orient)
    print(" feat equal", np.array_equal(f[0].cpu().numpy().view(np.uint16), ref), flush=True)
    out = seg.segment_device(d); torch.cuda.synchronize(); print(" device ok", flush=True)
    out = seg.segment_batch(imgs); torch.cuda.synchronize(); print(" batch ok", flush=True)
    out = seg.segment_batch(imgs); torch.cuda.synchronize(); print(" batch replay ok", flush=True)
print("done")
