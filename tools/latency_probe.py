import time, numpy as np, torch, sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gabor_color_image_segmentation_amd import Segmenter, segment
from gabor_color_image_segmentation_amd.synthetic import synthetic_batch
imgs = synthetic_batch(4, 321, 481, seed=1)
seg = Segmenter()
for _ in range(3): seg(imgs[0])
t=time.perf_counter()
for i in range(20): seg(imgs[i%4])
print("segment(img) host->host: %.3f ms" % ((time.perf_counter()-t)/20*1e3))
d = torch.from_numpy(imgs[:1]).cuda()
for _ in range(3): seg.segment_device(d)
torch.cuda.synchronize(); t=time.perf_counter()
for i in range(20): seg.segment_device(d)
torch.cuda.synchronize()
print("segment_device B=1: %.3f ms" % ((time.perf_counter()-t)/20*1e3))
# cpu-side launch cost only
t=time.perf_counter()
for i in range(20): seg.segment_device(d)
t1=time.perf_counter()-t; torch.cuda.synchronize()
print("  host enqueue time per call: %.3f ms" % (t1/20*1e3))
