"""The quality gate: BSD500 `val` boundary P / R / F of the slot's occupants, scored by the reference's own metrics
class (tests/golden/make_bsd_val_scores.py -> bsd_val_scores.json: 100 ids x {SPEC v2, round-1 SPEC, float64
full-resolution skimage bank, SLIC}) and, for the first 24 ids, the decoded images + the oracle's label maps
(bsd_val_images.npz). CPU side: the golden is self-consistent, the scoring mirror reproduces the reference's floats on
24 more label maps, and the C oracle reproduces the stored maps. The GPU side is tests/test_gpu_golden.py."""
import json
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def val():
    return json.load(open(os.path.join(GOLD, "bsd_val_scores.json"))), np.load(os.path.join(GOLD, "bsd_val_images.npz"))


def test_val_golden_is_the_whole_split_and_its_means_are_the_means(val):
    doc, _ = val
    ids = doc["ids"]
    assert len(ids) == 100 and ids == sorted(ids) and set(doc["per_id"]) == set(ids)
    for name in ("v2", "r1", "float", "slic"):
        for key in ("recall", "precision", "fmeasure"):
            assert doc["mean"][name][key] == float(np.mean([doc["per_id"][i][name][key] for i in ids]))
        for i in ids:
            s = doc["per_id"][i][name]
            assert s["fmeasure"] == 2.0 * s["precision"] * s["recall"] / (s["precision"] + s["recall"])


def test_spec_v2_scores_within_the_stated_distance_of_the_float_full_resolution_bank(val):
    """The gate DESIGN.md §7 quotes: the octave-pyramid fixed-point SPEC neither gains nor loses boundary F against a
    float64 full-resolution skimage Gabor bank with the same Lloyd schedule (mean over the 100 val ids), and is not
    below round 1's full-resolution 15x15 SPEC."""
    doc, _ = val
    m = doc["mean"]
    assert abs(m["v2"]["fmeasure"] - m["float"]["fmeasure"]) < 0.005
    assert m["v2"]["fmeasure"] >= m["r1"]["fmeasure"] - 0.002
    worst = max(abs(doc["per_id"][i]["v2"]["fmeasure"] - doc["per_id"][i]["float"]["fmeasure"]) for i in doc["ids"])
    assert worst < 0.13          # single images move (k-means basins), the split mean does not


def test_scoring_mirror_reproduces_the_reference_floats_on_the_stored_val_maps(val):
    """evaluate.metrics (the mirror of BSD_metrics/metrics.py) on the 24 stored oracle label maps and the packed ground
    truth == the numbers the reference class printed for them."""
    from gabor_color_image_segmentation_amd.evaluate import metrics
    from gabor_color_image_segmentation_amd.groundtruth import PackedTruth
    doc, pack = val
    pt = PackedTruth(os.path.join(GOLD, "bsd500_truth.npz"))
    for i in pack["ids"]:
        i = str(i)
        m = metrics(pack["img_" + i], pack["labels_" + i].astype(np.int32), pt[i])
        m.set_metrics()
        got, ref = m.get_metrics(), doc["per_id"][i]["v2"]
        assert got["regions"] == ref["regions"]
        for key in ("recall", "precision", "density"):
            assert got[key] == ref[key], (i, key)
        for key in ("underseg", "undersegNP", "compactness"):
            assert abs(got[key] - ref[key]) <= 1e-12, (i, key)


def test_c_oracle_reproduces_stored_val_label_maps(val, built):
    from oracle import c_oracle, spec_oracle
    _, pack = val
    tapq, shift = spec_oracle.bank()
    for i in (str(pack["ids"][0]), str(pack["ids"][3])):               # one portrait, one landscape
        lab = c_oracle.segment_batch(pack["img_" + i][None], tapq.astype(np.int16), shift, 6)[0]
        assert np.array_equal(lab, pack["labels_" + i])
