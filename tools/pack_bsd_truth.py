#!/usr/bin/env python3
"""Pack BSD500 ground truth into ONE .npz, so that a whole-dataset evaluation needs neither .mat
parsing nor the per-id scan of every split directory (/root/reference/BSD_metrics/groundtruth.py:33-50).

Run in the container that holds the reference, with the reference's own loader doing the reading (only data leaves the
reference tree, no source):

    cd /root/reference/BSD_metrics && PYTHONDONTWRITEBYTECODE=1 /opt/conda/bin/python3.9 \\
        /root/repo/tools/pack_bsd_truth.py OUT.npz [--ids 24 | --ids all | --ids 100075,100080] [--split train]

Format (read by gabor_color_image_segmentation_amd.groundtruth.PackedTruth):
    ids   (N,)   str      image ids, sorted
    hw    (N,2)  int32    map height, width (321x481 or 481x321)
    first (N+1,) int64    annotator maps of image i are maps first[i] .. first[i+1]-1 (4-9 per image)
    offs  (T+1,) int64    map t is data[offs[t]:offs[t+1]] reshaped to hw of its image
    data  (sum,) uint8    the `Segmentation` label maps (groundtruth.py:22-26: img[0][0][0]), uint16 if a label exceeds 255
All 500 ids: 2 696 maps, ~416 MB raw, a few tens of MB compressed.
"""
import os
import sys

import numpy as np

sys.path.insert(0, '.')
from groundtruth import get_segment_from_filename   # noqa: E402  (the reference's loader, groundtruth.py:33)

out = sys.argv[1]
want, split = "24", None
for i, a in enumerate(sys.argv):
    if a == "--ids":
        want = sys.argv[i + 1]
    if a == "--split":
        split = sys.argv[i + 1]
splits = [split] if split else sorted(os.listdir("data/truth"))
all_ids = sorted({f[:-4] for s in splits for f in os.listdir(os.path.join("data/truth", s)) if f.endswith(".mat")})
if want == "all":
    ids = all_ids
elif want.isdigit() and len(want) < 4:
    n = int(want)
    must = [i for i in ("100075", "100080", "100098") if i in all_ids]         # the fixture images stay in every pack
    rest = [i for i in all_ids if i not in must]
    step = max(1, len(rest) // max(1, n - len(must)))
    ids = sorted(must + rest[::step][:n - len(must)])
else:
    ids = sorted(want.split(","))
hw, first, offs, chunks = [], [0], [0], []
wide = False
for i in ids:
    segs = get_segment_from_filename(i)
    assert len(segs) > 0, i
    hw.append(segs[0].shape)
    for s in segs:
        assert s.shape == segs[0].shape and s.min() >= 0
        wide = wide or int(s.max()) > 255
        chunks.append(np.ascontiguousarray(s).astype(np.uint16).ravel())
        offs.append(offs[-1] + s.size)
    first.append(first[-1] + len(segs))
data = np.concatenate(chunks)
if not wide:
    data = data.astype(np.uint8)
np.savez_compressed(out, ids=np.array(ids), hw=np.array(hw, np.int32), first=np.array(first, np.int64),
                    offs=np.array(offs, np.int64), data=data)
print(f"{out}: {len(ids)} ids, {first[-1]} maps, {data.nbytes / 1e6:.1f} MB raw ({data.dtype}), "
      f"{os.path.getsize(out) / 1e6:.2f} MB on disk")
