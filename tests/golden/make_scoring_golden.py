"""Step 3/3 (run under /opt/conda/bin/python3.9, cwd = /root/reference/BSD_metrics).

Imports the reference's own ``metrics`` class (BSD_metrics/metrics.py:18) and records
recall (metrics.py:58-74) and precision (metrics.py:77-96) for several label maps per
fixture image: the oracle's Gabor+k-means map, a hand-made 2-region map, a 16x16 block
grid, and scikit-image SLIC (the slot's present occupant, script.py:30). Also the four
"next row" metrics (density, undersegmentation x2, compactness: metrics.py:102-201).
These numbers pin the build's scipy restatement of the scoring.
"""
import sys
import json
import numpy as np

sys.path.insert(0, '.')
from metrics import metrics            # noqa: E402  (the reference class)
from skimage.segmentation import slic  # noqa: E402

golden_dir = sys.argv[1]
inp = np.load(golden_dir + '/bsd_inputs.npz')
path = np.load(golden_dir + '/path_golden.npz')
res = {}
maps = {}
for i in inp['ids']:
    i = str(i)
    img = inp['img_' + i]
    segs = [inp['seg_%s_%d' % (i, a)] for a in range(int(inp['nseg_' + i]))]
    h, w = img.shape[:2]
    cand = {
        'oracle': path['labels_' + i].astype(np.int32),
        'halves': (np.arange(w)[None, :] >= w // 2).astype(np.int32) * np.ones((h, 1), np.int32),
        'blocks': ((np.arange(h)[:, None] // 16) * ((w + 15) // 16) + np.arange(w)[None, :] // 16).astype(np.int32),
        'slic': slic(img, n_segments=300, compactness=10.0).astype(np.int32),
    }
    for name, lab in cand.items():
        m = metrics(img, lab, segs)
        m.set_metrics()
        g = m.get_metrics()
        res[i + '/' + name] = {k: float(v) for k, v in g.items()}
        print(i, name, res[i + '/' + name])
    maps['slic_' + i] = cand['slic'].astype(np.uint16)
np.savez_compressed(golden_dir + '/scoring_maps.npz', **maps)
json.dump(res, open(golden_dir + '/scoring_golden.json', 'w'), indent=1, sort_keys=True)
