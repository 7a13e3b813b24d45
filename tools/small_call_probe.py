"""segment(img) for one 481x321 image, 30 calls (graph replay path); run under rocprofv3 --kernel-trace and show the last
step with tools/small_step_timeline.py."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gabor_color_image_segmentation_amd import Segmenter
from gabor_color_image_segmentation_amd.synthetic import synthetic_batch
imgs = synthetic_batch(4, 321, 481, seed=1)
seg = Segmenter()
for _ in range(5):
    seg(imgs[0])
t = time.perf_counter()
for i in range(30):
    seg(imgs[i % 4])
print("segment(img) host->host: %.3f ms" % ((time.perf_counter() - t) / 30 * 1e3))
