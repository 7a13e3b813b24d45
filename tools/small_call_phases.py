"""Host-side phases of one segment(img) call on the graph path (microseconds, median of 40 calls)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gabor_color_image_segmentation_amd import Segmenter
from gabor_color_image_segmentation_amd.segmenter import _stage
from gabor_color_image_segmentation_amd.synthetic import synthetic_batch
imgs = synthetic_batch(4, 321, 481, seed=1)
seg = Segmenter()
for _ in range(5):
    seg(imgs[0])
ent = next(iter(seg._graphs.values()))
cur = torch.cuda.current_stream()
rows = []
for i in range(40):
    im = imgs[i % 4][None]
    T = [time.perf_counter()]
    _stage(ent["pin_in"], im); T.append(time.perf_counter())
    ent["dev_in"].copy_(ent["pin_in"], non_blocking=True); T.append(time.perf_counter())
    ent["graph"].replay(); T.append(time.perf_counter())
    res = torch.empty((1, 321, 481), dtype=torch.int32, pin_memory=True); T.append(time.perf_counter())
    res.copy_(ent["dev_out"], non_blocking=True); T.append(time.perf_counter())
    cur.synchronize(); T.append(time.perf_counter())
    out = res.numpy(); T.append(time.perf_counter())
    rows.append([(T[j + 1] - T[j]) * 1e6 for j in range(len(T) - 1)] + [(T[-1] - T[0]) * 1e6])
med = np.median(np.array(rows), axis=0)
print("stage %.1f  h2d-enqueue %.1f  graph-replay-enqueue %.1f  pinned-alloc %.1f  d2h-enqueue %.1f  synchronize %.1f  numpy %.1f  total %.1f us" % tuple(med))
# the device part alone: replay + sync
ts = []
for i in range(40):
    torch.cuda.synchronize(); t0 = time.perf_counter(); ent["graph"].replay(); cur.synchronize(); ts.append((time.perf_counter() - t0) * 1e6)
print("graph replay + synchronize alone: %.1f us" % np.median(ts))
