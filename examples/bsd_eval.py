#!/usr/bin/env python3
"""The reference's driver loop (/root/reference/BSD_metrics/script.py:19-38) with the MI355X segmenter
in the slot, on the packed BSD fixtures (no JPEG / .mat decoding, no /root/reference needed):

    load image -> labels = segment(img) -> load ground truth -> metrics -> print

Iterates an explicit sorted id list (script.py:21-22 takes os.listdir order, which is not stable).
`--gpu-scoring`: the label map never leaves the device; gcs_boundary_counts / gcs_region_counts produce the integer
tables and the same numbers are printed (evaluate_gpu.all_scores_device).
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from gabor_color_image_segmentation_amd import segment                      # noqa: E402  (script.py:11)
from gabor_color_image_segmentation_amd.evaluate import metrics             # noqa: E402  (script.py:14)
from gabor_color_image_segmentation_amd.groundtruth import load_packed      # noqa: E402  (script.py:13)

if __name__ == '__main__':
    data = load_packed(os.path.join(ROOT, "tests", "golden", "bsd_inputs.npz"))
    for name in sorted(data):
        img, segments = data[name]                       # script.py:25, :33
        print("Processing image " + name)
        if "--gpu-scoring" in sys.argv:
            import torch
            from gabor_color_image_segmentation_amd import Segmenter
            from gabor_color_image_segmentation_amd.evaluate_gpu import all_scores_device
            seg = Segmenter()
            s = all_scores_device(seg.segment_device(torch.from_numpy(img[None]).cuda())[0], segments)
            print("Regions: %d Recall: %r Precision: %r F-measure: %r Undersegmentation: %r Undersegmentation (NP) %r "
                  "Compactness %r Density %r" % (s["regions"], s["recall"], s["precision"], s["fmeasure"], s["underseg"],
                                                 s["undersegNP"], s["compactness"], s["density"]))
            continue
        labels = segment(img)                            # script.py:30 — the slot
        m = metrics(img, labels, segments)               # script.py:36
        m.set_metrics()
        m.display_metrics()
