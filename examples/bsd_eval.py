#!/usr/bin/env python3
"""The reference's driver loop (/root/reference/BSD_metrics/script.py:19-38) with the MI355X segmenter
in the slot, on the packed BSD fixtures (no JPEG / .mat decoding, no /root/reference needed):

    load image -> labels = segment(img) -> load ground truth -> metrics -> print

Iterates an explicit sorted id list (script.py:21-22 takes os.listdir order, which is not stable).
`--gpu-scoring`: the label map never leaves the device; gcs_boundary_counts / gcs_region_counts produce the integer
tables and the same numbers are printed (evaluate_gpu.all_scores_device).
`--val`: the data-set form of the same loop on the 24 packed images of the BSD500 val split: batches per image shape through
the device path, the batched GPU scorer against the packed ground truth of all 500 ids, mean recall / precision / F beside the
numbers the reference's own metrics class gave for the same label maps (tests/golden/bsd_val_scores.json).
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from gabor_color_image_segmentation_amd import segment                      # noqa: E402  (script.py:11)
from gabor_color_image_segmentation_amd.evaluate import metrics             # noqa: E402  (script.py:14)
from gabor_color_image_segmentation_amd.groundtruth import load_packed      # noqa: E402  (script.py:13)

def val_split():
    import json
    import numpy as np
    import torch
    from gabor_color_image_segmentation_amd import Segmenter
    from gabor_color_image_segmentation_amd.evaluate_gpu import all_scores_batch_device
    from gabor_color_image_segmentation_amd.groundtruth import PackedTruth
    gold = os.path.join(ROOT, "tests", "golden")
    pack = np.load(os.path.join(gold, "bsd_val_images.npz"))
    ref = json.load(open(os.path.join(gold, "bsd_val_scores.json")))["per_id"]
    truth = PackedTruth(os.path.join(gold, "bsd500_truth.npz"))
    seg = Segmenter()
    ids = [str(i) for i in pack["ids"]]
    rows = {}
    for shape in sorted({pack["img_" + i].shape[:2] for i in ids}):
        group = [i for i in ids if pack["img_" + i].shape[:2] == shape]
        labels = seg.segment_device(torch.from_numpy(np.stack([pack["img_" + i] for i in group])).cuda())
        # ground truth resident on the device in the scorer's form (prepared once per shape group; a data-set loop keeps it)
        for i, s in zip(group, all_scores_batch_device(labels, truth.to_device(group), n_segments=seg.k)):
            rows[i] = s
            print("%-8s Regions: %d Recall: %.4f Precision: %.4f F-measure: %.4f   (reference class: %.4f)"
                  % (i, s["regions"], s["recall"], s["precision"], s["fmeasure"], ref[i]["v2"]["fmeasure"]))
    for key in ("recall", "precision", "fmeasure"):
        print("mean %-9s %.4f   reference class on the same maps: %.4f" % (
            key, float(np.mean([rows[i][key] for i in ids])), float(np.mean([ref[i]["v2"][key] for i in ids]))))


if __name__ == '__main__':
    if "--val" in sys.argv:
        val_split()
        sys.exit(0)
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
