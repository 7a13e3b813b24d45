"""Step 1/3 (run under /opt/conda/bin/python3.9, cwd = /root/reference/BSD_metrics).

Decodes a fixed, sorted list of BSD500 ids with the reference's own loaders and stores
the *decoded arrays* (so no JPEG / .mat decoder is needed on the GPU box):
  image:  skimage.io.imread                      (BSD_metrics/script.py:25)
  truth:  groundtruth.get_segment_from_filename  (BSD_metrics/groundtruth.py:33-50)
Only data leaves the reference tree; no reference source is copied.
"""
import sys
import numpy as np

sys.path.insert(0, '.')
from skimage.io import imread            # noqa: E402
from groundtruth import get_segment_from_filename   # noqa: E402

IDS = ['100075', '100080', '100098']     # landscape, portrait, landscape (sorted listdir order)
out = {}
for i in IDS:
    img = imread('data/Berkeley/train/' + i + '.jpg')
    segs = get_segment_from_filename(i)
    assert img.dtype == np.uint8 and img.ndim == 3 and len(segs) > 0
    out['img_' + i] = img
    out['nseg_' + i] = np.int64(len(segs))
    for a, s in enumerate(segs):
        out['seg_%s_%d' % (i, a)] = s.astype(np.uint16)
np.savez_compressed(sys.argv[1], ids=np.array(IDS), **out)
print('wrote', sys.argv[1], {k: v.shape for k, v in out.items() if k.startswith('img_')})
