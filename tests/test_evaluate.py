"""The scoring mirror against numbers produced by the reference's own metrics class
(tests/golden/make_scoring_golden.py ran /root/reference/BSD_metrics/metrics.py)."""
import json
import os
import numpy as np
import pytest

from gabor_color_image_segmentation_amd.evaluate import metrics, boundary_scores, find_boundaries
from oracle import spec_oracle as so

GOLD = os.path.join(os.path.dirname(__file__), "golden")
INP = np.load(os.path.join(GOLD, "bsd_inputs.npz"))
PATH = np.load(os.path.join(GOLD, "path_golden.npz"))
MAPS = np.load(os.path.join(GOLD, "scoring_maps.npz"))
SCORES = json.load(open(os.path.join(GOLD, "scoring_golden.json")))


def _segs(i):
    return [INP["seg_%s_%d" % (i, a)] for a in range(int(INP["nseg_" + i]))]


def _label_map(i, name):
    h, w = INP["img_" + i].shape[:2]
    if name == "oracle":
        return PATH["labels_" + i].astype(np.int32)
    if name == "halves":
        return (np.arange(w)[None, :] >= w // 2).astype(np.int32) * np.ones((h, 1), np.int32)
    if name == "blocks":
        return ((np.arange(h)[:, None] // 16) * ((w + 15) // 16) + np.arange(w)[None, :] // 16).astype(np.int32)
    return MAPS["slic_" + i].astype(np.int32)


@pytest.mark.parametrize("key", sorted(SCORES))
def test_all_metrics_equal_the_reference(key):
    i, name = key.split("/")
    m = metrics(INP["img_" + i], _label_map(i, name), _segs(i))
    m.set_metrics()
    got, ref = m.get_metrics(), SCORES[key]
    assert got["regions"] == ref["regions"]
    for k in ("recall", "precision", "density"):
        assert got[k] == ref[k], k                     # same integer counts, same float divisions
    for k in ("underseg", "undersegNP", "compactness"):
        assert got[k] == pytest.approx(ref[k], rel=1e-12), k
    p, r = ref["precision"], ref["recall"]
    assert got["fmeasure"] == pytest.approx(2 * p * r / (p + r), rel=1e-15)


def test_boundary_scores_match_oracle_restatement():
    i = "100080"
    lab = _label_map(i, "oracle")
    s = boundary_scores(lab, _segs(i))
    r, p = so.boundary_recall_precision(lab, _segs(i))
    assert s["recall"] == r and s["precision"] == p and s["fmeasure"] == so.fmeasure(r, p)


def test_label_permutation_does_not_change_boundary_scores():
    i = "100075"
    lab = _label_map(i, "oracle")
    perm = np.array([3, 7, 1, 0, 6, 2, 5, 4])
    assert boundary_scores(lab, _segs(i)) == boundary_scores(perm[lab], _segs(i))


def test_constant_label_map_divides_by_zero_like_the_reference():
    """metrics.py:94 divides by the number of boundary pixels: a one-label map raises."""
    i = "100075"
    h, w = INP["img_" + i].shape[:2]
    m = metrics(None, np.zeros((h, w), np.int32), _segs(i))
    m.set_boundary_recall()
    assert m.recall == 0.0
    with pytest.raises(ZeroDivisionError):
        m.set_boundary_precision()


def test_empty_truth_list_divides_by_zero_like_the_reference():
    m = metrics(None, np.zeros((8, 8), np.int32), [])
    with pytest.raises(ZeroDivisionError):
        m.set_boundary_recall()                        # metrics.py:74


def test_find_boundaries_is_thick_and_ignores_the_image_border():
    lab = np.zeros((6, 6), np.int32)
    lab[:, 3:] = 1
    b = find_boundaries(lab)
    assert b[:, 2].all() and b[:, 3].all() and b.sum() == 12


def test_packed_groundtruth_reader():
    from gabor_color_image_segmentation_amd.groundtruth import load_packed
    data = load_packed(os.path.join(GOLD, "bsd_inputs.npz"))
    assert sorted(data) == ["100075", "100080", "100098"]
    img, segs = data["100080"]
    assert img.shape == (481, 321, 3) and img.dtype == np.uint8 and len(segs) == int(INP["nseg_100080"])
    assert all(s.shape == (481, 321) for s in segs)


@pytest.mark.skipif(not os.path.isdir("/root/reference/BSD_metrics/data/truth"), reason="reference data not mounted")
def test_mat_loader_mirror_matches_the_fixture():
    """groundtruth.get_segment_from_filename mirror on the reference's own .mat files (this container only)."""
    from gabor_color_image_segmentation_amd.groundtruth import get_segment_from_filename
    segs = get_segment_from_filename("100080", path="/root/reference/BSD_metrics/data/truth/")
    assert len(segs) == int(INP["nseg_100080"])
    for a, s in enumerate(segs):
        assert np.array_equal(s, INP["seg_100080_%d" % a])


def test_whole_dataset_truth_pack_matches_the_fixtures_and_the_mat_loader():
    """tests/golden/bsd500_truth.npz (tools/pack_bsd_truth.py: the reference's own loader over all 500 ids) holds the same
    maps as the per-image fixtures, and as the .mat mirror where the reference tree is present."""
    from gabor_color_image_segmentation_amd.groundtruth import PackedTruth, get_segment_from_filename
    pt = PackedTruth(os.path.join(GOLD, "bsd500_truth.npz"))
    assert len(pt) == 500 and int(pt.first[-1]) == 2696
    assert sorted(pt.shape(i) for i in pt.ids)[0] == (321, 481) and sorted(pt.shape(i) for i in pt.ids)[-1] == (481, 321)
    assert min(pt.n_annotators(i) for i in pt.ids) >= 4 and max(pt.n_annotators(i) for i in pt.ids) <= 9
    for i in INP["ids"]:
        i = str(i)
        got = pt[i]
        assert len(got) == int(INP["nseg_" + i])
        for a, m in enumerate(got):
            assert m.dtype == np.uint16 and np.array_equal(m, INP["seg_%s_%d" % (i, a)])
    truth, first, img_of, n_truth = pt.stack(["100075", "100098"])
    assert truth.shape[1:] == (321, 481) and first.tolist() == [0, pt.n_annotators("100075"), truth.shape[0]]
    assert img_of.tolist() == [0] * first[1] + [1] * (truth.shape[0] - first[1]) and n_truth[0] == int(truth[0].max()) + 1
    with pytest.raises(ValueError):
        pt.stack(["100075", "100080"])                     # landscape + portrait
    ref = "/root/reference/BSD_metrics/data/truth/"
    if os.path.isdir(ref):
        for i in ("2092", "97010", "100080"):
            want = get_segment_from_filename(i, ref)
            assert len(want) == len(pt[i]) and all(np.array_equal(x, y) for x, y in zip(want, pt[i]))
