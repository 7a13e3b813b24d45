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
    resident_truth = {}
    for ids in (["100075", "100098"], ["100080"]):                       # one batch per image shape
        for name in ("oracle", "halves", "blocks", "slic"):
            labs = np.stack([_maps(i, inp, path, maps)[name] for i in ids])
            dev = torch.from_numpy(np.ascontiguousarray(labs)).cuda()
            got = all_scores_batch_device(dev, *pt.stack(ids))
            # the same through the RESIDENT ground truth (bit planes + uint8 maps prepared once: gcs_truth_prepare,
            # gcs_boundary_counts_resident, gcs_region_counts_batch_u8): identical numbers, float for float
            resident = resident_truth.setdefault(tuple(ids), pt.to_device(ids))
            assert all_scores_batch_device(dev, resident) == got, (ids, name)
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
        resident = pt.to_device(group)                                    # prepared ONCE per group, scored three times
        for v in range(3):
            labs = np.stack([_label_candidates(h, w, seed=int(i))[v] for i in group])
            dev = torch.from_numpy(labs).cuda()
            got = all_scores_batch_device(dev, *stack)
            assert all_scores_batch_device(dev, resident) == got, (group[0], v)      # resident truth: the same floats
            assert all_scores_batch_device(dev, resident, n_segments=int(labs.max()) + 1) == got
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


def test_resident_truth_bit_planes_equal_scipy(built):
    """gcs_truth_prepare against the scipy restatement of the reference's stencils (evaluate.find_boundaries /
    _dilate(., 5): metrics.py:49, :69, :93): every bit of both planes, the boundary counts and the uint8 maps, on shapes whose width is
    and is not a multiple of 64 and whose boundaries touch every border."""
    import torch
    from gabor_color_image_segmentation_amd import evaluate as ev
    from gabor_color_image_segmentation_amd.evaluate_gpu import DeviceTruth
    rng = np.random.default_rng(3)
    for h, w in ((321, 481), (37, 64), (5, 130), (70, 63)):
        maps = []
        for t in range(3):
            pts = rng.uniform(0, 1, (6 + t, 2)) * [h, w]
            yy, xx = np.mgrid[:h, :w]
            maps.append((np.argmin((yy[..., None] - pts[:, 0]) ** 2 + (xx[..., None] - pts[:, 1]) ** 2, axis=2) + 1).astype(np.uint16))
        maps[1][0, :] = 9; maps[1][:, -1] = 11; maps[2][-1, -1] = 7         # boundaries on the borders and in the last bit
        truth = np.stack(maps)
        dt = DeviceTruth(truth, [0, 2, 3], [0, 0, 1], [int(m.max()) + 1 for m in maps])
        wp = (w + 63) // 64
        planes = dt.planes.cpu().numpy().view(np.uint64).reshape(2, 3, h, wp)
        bits = np.unpackbits(planes.view(np.uint8), axis=-1, bitorder="little").reshape(2, 3, h, wp * 64)
        assert not bits[..., w:].any()                                     # nothing beyond the image's last column
        for t in range(3):
            bd = ev.find_boundaries(truth[t])
            assert np.array_equal(bits[0, t, :, :w].astype(bool), bd), (h, w, t)
            assert np.array_equal(bits[1, t, :, :w].astype(bool), ev._dilate(bd, 5)), (h, w, t)
            assert int(dt.bd_counts[t].item()) == int(bd.sum())
        assert dt.u8 and np.array_equal(dt.maps.cpu().numpy(), truth.astype(np.uint8))


def test_the_fused_resident_scorer_equals_its_three_parts(built):
    """gcs_score_batch_resident (six launches: zero | label bit planes + maxima | their dilation | boundary counts | region tables | their reduction) against the three entry points it
    fuses - gcs_boundary_counts_resident (dilated label planes from their own launch), gcs_region_counts_batch_u8 and
    gcs_region_reduce - and gcs_region_reduce against NumPy on the tables: every output word equal."""
    import torch
    from gabor_color_image_segmentation_amd import _lib
    from gabor_color_image_segmentation_amd.groundtruth import PackedTruth
    lib = _lib.load()
    pt = PackedTruth(os.path.join(GOLD, "bsd500_truth.npz"))
    group = [i for i in pt.ids if pt.shape(i) == (481, 321)][:5]
    dt = pt.to_device(group)
    b, t, h, w, n_seg = dt.b, dt.t, dt.h, dt.w, 9
    labs = torch.from_numpy(np.stack([_label_candidates(h, w, seed=int(i))[2] for i in group])).cuda()
    stream = torch.cuda.current_stream().cuda_stream
    z = lambda n, dtype: torch.full((n,), -1, dtype=dtype, device="cuda")               # outputs start dirty: nothing may rely on zeros
    scratch = torch.empty(lib.gcs_bit_planes_bytes(b, h, w), dtype=torch.uint8, device="cuda")
    hist, counts, smax = z(t * n_seg * dt.stride, torch.int32), z(b + 3 * t, torch.int64), z(b, torch.int32)
    area, perim, under, under_np = z(b * n_seg, torch.int32), z(b * n_seg, torch.int32), z(t, torch.int64), z(t, torch.int64)
    _lib.check(lib.gcs_score_batch_resident(labs.data_ptr(), dt.planes.data_ptr(), dt.bd_counts.data_ptr(), dt.maps.data_ptr(), 1,
                                            dt.first_d.data_ptr(), dt.img_of_d.data_ptr(), b, t, dt.a_max, h, w, n_seg, dt.stride,
                                            scratch.data_ptr(), hist.data_ptr(), counts.data_ptr(), smax.data_ptr(), area.data_ptr(),
                                            perim.data_ptr(), under.data_ptr(), under_np.data_ptr(), stream), "fused")
    hist2, counts2, smax2 = z(t * n_seg * dt.stride, torch.int32), z(b + 3 * t, torch.int64), z(b, torch.int32)
    area2, perim2, under2, under_np2 = z(b * n_seg, torch.int32), z(b * n_seg, torch.int32), z(t, torch.int64), z(t, torch.int64)
    _lib.check(lib.gcs_boundary_counts_resident(labs.data_ptr(), dt.planes.data_ptr(), dt.bd_counts.data_ptr(), dt.img_of_d.data_ptr(),
                                                b, t, h, w, scratch.data_ptr(), counts2.data_ptr(), smax2.data_ptr(), stream), "counts")
    _lib.check(lib.gcs_region_counts_batch_u8(labs.data_ptr(), dt.maps.data_ptr(), dt.first_d.data_ptr(), b, t, dt.a_max, h, w, n_seg,
                                              dt.stride, hist2.data_ptr(), area2.data_ptr(), perim2.data_ptr(), stream), "regions")
    _lib.check(lib.gcs_region_reduce(hist2.data_ptr(), area2.data_ptr(), dt.img_of_d.data_ptr(), t, n_seg, dt.stride,
                                     under2.data_ptr(), under_np2.data_ptr(), stream), "reduce")
    for a, c in ((hist, hist2), (counts, counts2), (smax, smax2), (area, area2), (perim, perim2), (under, under2), (under_np, under_np2)):
        assert torch.equal(a, c)
    assert smax.cpu().tolist() == [int(l.max()) for l in labs.cpu()]
    hh = hist.cpu().numpy().reshape(t, n_seg, dt.stride).astype(np.int64)
    ar = area.cpu().numpy().reshape(b, n_seg).astype(np.int64)[dt.img_of]
    assert np.array_equal(under.cpu().numpy(), (ar - hh.max(axis=2)).sum(axis=1))
    assert np.array_equal(under_np.cpu().numpy(), np.minimum(hh, hh.sum(axis=2, keepdims=True) - hh).sum(axis=(1, 2)))
