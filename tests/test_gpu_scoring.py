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


def _label_candidates(h, w, seed):
    """Label maps that need no image: a 16x16 block grid, two halves and a seeded Voronoi partition (8 cells)."""
    rng = np.random.default_rng(seed)
    pts = rng.uniform(0, 1, (8, 2)) * [h, w]
    yy, xx = np.mgrid[:h, :w]
    vor = np.argmin((yy[..., None] - pts[:, 0]) ** 2 + (xx[..., None] - pts[:, 1]) ** 2, axis=2).astype(np.int32)
    return [((np.arange(h)[:, None] // 16) * ((w + 15) // 16) + np.arange(w)[None, :] // 16).astype(np.int32),
            ((np.arange(w)[None, :] >= w // 2).astype(np.int32) * np.ones((h, 1), np.int32)), vor]


def test_batched_gpu_scorer_equals_the_reference_numbers_on_the_golden_maps(built):
    """The 12 golden maps (3 BSD ids x {oracle, halves, blocks, SLIC}) through gcs_boundary_counts_batch /
    gcs_region_counts_batch (ragged annotator stack, one launch per table for the whole batch) give the reference's own
    numbers (tests/golden/scoring_golden.json, produced by /root/reference/BSD_metrics/metrics.py)."""
    import torch
    from gabor_color_image_segmentation_amd.evaluate_gpu import all_scores_batch_device
    from gabor_color_image_segmentation_amd.groundtruth import PackedTruth
    inp = np.load(os.path.join(GOLD, "bsd_inputs.npz"))
    path = np.load(os.path.join(GOLD, "path_golden.npz"))
    maps = np.load(os.path.join(GOLD, "scoring_maps.npz"))
    scores = json.load(open(os.path.join(GOLD, "scoring_golden.json")))
    pt = PackedTruth(os.path.join(GOLD, "bsd500_truth.npz"))
    for ids in (["100075", "100098"], ["100080"]):                       # one batch per image shape
        for name in ("oracle", "halves", "blocks", "slic"):
            labs = np.stack([_maps(i, inp, path, maps)[name] for i in ids])
            got = all_scores_batch_device(torch.from_numpy(np.ascontiguousarray(labs)).cuda(), *pt.stack(ids))
            for i, g in zip(ids, got):
                ref = scores[i + "/" + name]
                assert g["regions"] == ref["regions"], (i, name)
                for key in ("recall", "precision", "underseg", "undersegNP", "density"):
                    assert g[key] == ref[key], (i, name, key, g[key], ref[key])
                assert abs(g["compactness"] - ref["compactness"]) <= 1e-15 * max(1.0, abs(ref["compactness"])), (i, name)


def test_batched_gpu_scorer_on_sixty_further_ids_equals_the_host_mirror(built):
    """Whole-dataset ingest: 60 more BSD ids straight from the packed ground truth (40 landscape,
    20 portrait; 4-9 annotators each), three label maps per id, scored in batches of 20 images by the batched GPU scorer
    and, one image at a time, by the host mirror of /root/reference/BSD_metrics/metrics.py (evaluate.metrics, itself pinned
    to the reference's numbers) and by the single-image GPU scorer."""
    import torch
    from gabor_color_image_segmentation_amd.evaluate import metrics
    from gabor_color_image_segmentation_amd.evaluate_gpu import all_scores_batch_device, all_scores_device
    from gabor_color_image_segmentation_amd.groundtruth import PackedTruth
    pt = PackedTruth(os.path.join(GOLD, "bsd500_truth.npz"))
    land = [i for i in pt.ids if pt.shape(i) == (321, 481) and i not in ("100075", "100098")][::8][:40]
    port = [i for i in pt.ids if pt.shape(i) == (481, 321) and i != "100080"][::7][:20]
    assert len(land) == 40 and len(port) == 20
    checked = 0
    for group in (land[:20], land[20:], port):
        h, w = pt.shape(group[0])
        stack = pt.stack(group)
        for v in range(3):
            labs = np.stack([_label_candidates(h, w, seed=int(i))[v] for i in group])
            dev = torch.from_numpy(labs).cuda()
            got = all_scores_batch_device(dev, *stack)
            for b, i in enumerate(group):
                if (b + v) % 5 == 0:                                   # the host loops are slow: every 5th (id, map) pair
                    m = metrics(np.zeros((h, w, 3), np.uint8), labs[b], pt[i])
                    m.set_metrics()
                    ref = m.get_metrics()
                    assert got[b]["regions"] == ref["regions"]
                    for key in ("recall", "precision", "fmeasure", "density"):
                        assert got[b][key] == ref[key], (i, v, key)
                    for key in ("underseg", "undersegNP", "compactness"):
                        assert abs(got[b][key] - ref[key]) <= 1e-12, (i, v, key)
                    checked += 1
                if b % 7 == 0:
                    assert got[b] == all_scores_device(dev[b], pt[i]), (i, v)
    assert checked >= 36
