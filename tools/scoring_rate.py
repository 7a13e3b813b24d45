#!/usr/bin/env python3
"""Throughput of the batched GPU scorer (gcs_boundary_counts_batch + gcs_region_counts_batch behind
evaluate_gpu.all_scores_batch_device on resident ground truth, evaluate_gpu.DeviceTruth) and of the segment + score loop (`examples/bsd_eval.py --val`) on the 24 packed BSD500
val images: label maps stay on the device, ground truth comes from the 500-id pack. Prints one JSON line; `bench.py` embeds
the same figures (`scoring`)."""
import json, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gabor_color_image_segmentation_amd import Segmenter
from gabor_color_image_segmentation_amd.evaluate_gpu import submit_scores_batch_resident
from gabor_color_image_segmentation_amd.groundtruth import PackedTruth


def measure(seg=None, reps=5):
    gold = os.path.join(ROOT, "tests", "golden")
    pack = np.load(os.path.join(gold, "bsd_val_images.npz"))
    truth = PackedTruth(os.path.join(gold, "bsd500_truth.npz"))
    seg = seg or Segmenter()
    ids = [str(i) for i in pack["ids"]]
    groups = []
    for shape in sorted({pack["img_" + i].shape[:2] for i in ids}):
        g = [i for i in ids if pack["img_" + i].shape[:2] == shape]
        imgs = torch.from_numpy(np.stack([pack["img_" + i] for i in g])).cuda()
        groups.append((g, imgs, truth.to_device(g)))      # resident ground truth: prepared once per shape group, kept on the device
    n_img = len(ids)
    n_maps = sum(st.t for _, _, st in groups)

    def seg_all():
        return [seg.segment_device(imgs) for _, imgs, _ in groups]

    def score_all(labs):
        # every group's kernels are enqueued before the first result is collected: the second group's kernels run under the
        # first group's host arithmetic
        pending = [submit_scores_batch_resident(l, st, n_segments=seg.k) for l, (_, _, st) in zip(labs, groups)]
        return [p.result() for p in pending]
    labs = seg_all(); score_all(labs); torch.cuda.synchronize()          # warm-up (workspaces, truth upload paths)
    t0 = time.perf_counter()
    for _ in range(reps):
        labs = seg_all()
    torch.cuda.synchronize()
    t_seg = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    for _ in range(reps):
        score_all(labs)                                                   # returns host floats: synchronises by itself
    t_score = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    for _ in range(reps):
        score_all(seg_all())
    t_loop = (time.perf_counter() - t0) / reps
    return dict(images=n_img, annotator_maps=n_maps, segment_ms=round(t_seg * 1e3, 3), score_ms=round(t_score * 1e3, 3),
                loop_ms=round(t_loop * 1e3, 3), scoring_images_s=round(n_img / t_score, 1),
                segment_images_s=round(n_img / t_seg, 1), bsd_eval_val_images_s=round(n_img / t_loop, 1),
                scoring_share_of_loop=round(t_score / t_loop, 3),
                note="24 BSD500 val images (two shape groups), device-resident label maps AND ground truth (annotator bit planes / "
                     "uint8 maps prepared once: gcs_truth_prepare), boundary + region kernels, one device-to-host copy per group, "
                     "host float arithmetic of metrics.py:58-201 included; per-image codebooks")


if __name__ == "__main__":
    print(json.dumps(measure()))
