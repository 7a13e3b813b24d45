"""GPU boundary evaluator against numbers from the reference's own metrics class."""
import json
import os
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _maps(i, inp, path, maps):
    h, w = inp["img_" + i].shape[:2]
    return {
        "oracle": path["labels_" + i].astype(np.int32),
        "halves": ((np.arange(w)[None, :] >= w // 2).astype(np.int32) * np.ones((h, 1), np.int32)),
        "blocks": ((np.arange(h)[:, None] // 16) * ((w + 15) // 16) + np.arange(w)[None, :] // 16).astype(np.int32),
        "slic": maps["slic_" + i].astype(np.int32),
    }


def test_gpu_recall_precision_equal_the_reference(built):
    import torch
    from gabor_color_image_segmentation_amd.evaluate_gpu import boundary_scores_device
    inp = np.load(os.path.join(GOLD, "bsd_inputs.npz"))
    path = np.load(os.path.join(GOLD, "path_golden.npz"))
    maps = np.load(os.path.join(GOLD, "scoring_maps.npz"))
    scores = json.load(open(os.path.join(GOLD, "scoring_golden.json")))
    for i in inp["ids"]:
        i = str(i)
        segs = [inp["seg_%s_%d" % (i, a)] for a in range(int(inp["nseg_" + i]))]
        for name, lab in _maps(i, inp, path, maps).items():
            got = boundary_scores_device(torch.from_numpy(np.ascontiguousarray(lab)).cuda(), segs)
            ref = scores[i + "/" + name]
            assert got["recall"] == ref["recall"] and got["precision"] == ref["precision"], (i, name)


def test_gpu_scores_of_gpu_segmentation_without_host_round_trip(built):
    import torch
    from gabor_color_image_segmentation_amd import Segmenter
    from gabor_color_image_segmentation_amd.evaluate import boundary_scores
    from gabor_color_image_segmentation_amd.evaluate_gpu import boundary_scores_device
    inp = np.load(os.path.join(GOLD, "bsd_inputs.npz"))
    i = "100080"
    segs = [inp["seg_%s_%d" % (i, a)] for a in range(int(inp["nseg_" + i]))]
    seg = Segmenter()
    lab = seg.segment_device(torch.from_numpy(inp["img_" + i][None]).cuda())[0]
    assert boundary_scores_device(lab, segs) == boundary_scores(lab.cpu().numpy(), segs)


def test_gpu_scoring_degenerate_maps_raise_like_the_reference(built):
    import torch
    from gabor_color_image_segmentation_amd.evaluate_gpu import boundary_scores_device
    truth = [np.tile(np.arange(16, dtype=np.uint16) // 8, (12, 1))]
    with pytest.raises(ZeroDivisionError):        # constant label map: metrics.py:94
        boundary_scores_device(torch.zeros((12, 16), dtype=torch.int32).cuda(), truth)
    with pytest.raises(ZeroDivisionError):        # no annotators: metrics.py:74
        boundary_scores_device(torch.zeros((12, 16), dtype=torch.int32).cuda(), [])


def test_gpu_region_metrics_equal_the_reference(built):
    """Undersegmentation (both formulas), compactness, density and the region count of all 12 golden maps: integer
    tables from gcs_region_counts, float arithmetic as metrics.py:128-146,194-201 -> the reference's own numbers."""
    import torch
    from gabor_color_image_segmentation_amd.evaluate_gpu import all_scores_device
    inp = np.load(os.path.join(GOLD, "bsd_inputs.npz"))
    path = np.load(os.path.join(GOLD, "path_golden.npz"))
    maps = np.load(os.path.join(GOLD, "scoring_maps.npz"))
    scores = json.load(open(os.path.join(GOLD, "scoring_golden.json")))
    for i in inp["ids"]:
        i = str(i)
        segs = [inp["seg_%s_%d" % (i, a)] for a in range(int(inp["nseg_" + i]))]
        for name, lab in _maps(i, inp, path, maps).items():
            got = all_scores_device(torch.from_numpy(np.ascontiguousarray(lab)).cuda(), segs)
            ref = scores[i + "/" + name]
            assert got["regions"] == ref["regions"], (i, name)
            for key in ("recall", "precision", "underseg", "undersegNP", "density"):
                assert got[key] == ref[key], (i, name, key, got[key], ref[key])
            assert abs(got["compactness"] - ref["compactness"]) <= 1e-15 * max(1.0, abs(ref["compactness"])), (i, name)


def test_gpu_region_tables_on_connected_regions(built):
    """Thousands of sparse regions (the global-atomics path) against the host mirror of the reference loops."""
    import torch
    from gabor_color_image_segmentation_amd import Segmenter
    from gabor_color_image_segmentation_amd.evaluate import metrics
    from gabor_color_image_segmentation_amd.evaluate_gpu import all_scores_device
    inp = np.load(os.path.join(GOLD, "bsd_inputs.npz"))
    i = "100080"
    segs = [inp["seg_%s_%d" % (i, a)] for a in range(int(inp["nseg_" + i]))]
    seg = Segmenter(connectivity=True)
    lab = seg.segment_device(torch.from_numpy(inp["img_" + i][None]).cuda())[0]
    got = all_scores_device(lab, segs)
    m = metrics(inp["img_" + i], lab.cpu().numpy(), segs)
    m.set_metrics()
    ref = m.get_metrics()
    assert got["regions"] == ref["regions"] and got["regions"] > 500
    for key in ("recall", "precision", "fmeasure", "density"):
        assert got[key] == ref[key], key
    for key in ("underseg", "undersegNP", "compactness"):
        assert abs(got[key] - ref[key]) <= 1e-12, key
