"""GPU at full BSD size: committed goldens, the reference-pinned scoring, and size-independent
properties at BASELINE.json's batch shape."""
import json
import os
import numpy as np
import pytest

from oracle import spec_oracle as so, c_oracle as co

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def seg(built):
    import torch
    assert torch.cuda.is_available()
    from gabor_color_image_segmentation_amd import Segmenter
    return Segmenter()


@pytest.fixture(scope="module")
def fixtures():
    return np.load(os.path.join(GOLD, "bsd_inputs.npz")), np.load(os.path.join(GOLD, "path_golden.npz"))


def test_bsd_images_labels_centroid_free_parity(seg, fixtures):
    """Both orientations ((321,481) and (481,321)), default bank/k/n_iter: bit-identical label maps."""
    inp, gold = fixtures
    for i in inp["ids"]:
        lab = seg(inp["img_" + str(i)])
        assert lab.dtype == np.int32 and lab.shape == inp["img_" + str(i)].shape[:2]
        assert np.array_equal(lab, gold["labels_" + str(i)]), f"{i}: {(lab != gold['labels_' + str(i)]).mean():.3%}"


def test_bsd_images_feature_goldens(seg, fixtures):
    import torch
    inp, gold = fixtures
    for i in inp["ids"][:2]:
        img = inp["img_" + str(i)]
        f = seg.features_device(torch.from_numpy(img[None]).cuda()).cpu().numpy().view(np.uint16)[0]
        assert np.array_equal(f.reshape(72, -1).astype(np.int64).sum(axis=1), gold["feat_sum_" + str(i)])
        assert np.array_equal(f.reshape(72, -1).max(axis=1), gold["feat_max_" + str(i)])
        for (y, x), pf in zip(gold["probe_yx_" + str(i)], gold["probe_feat_" + str(i)]):
            assert np.array_equal(f[:, y, x], pf)


def test_bsd_boundary_f_measure_identical_to_reference_scoring(seg, fixtures):
    """P / R from the reference's own metrics class on the oracle's label map
    (scoring_golden.json) == P / R of the GPU label map through the scoring mirror."""
    from gabor_color_image_segmentation_amd.evaluate import boundary_scores
    inp, _ = fixtures
    scores = json.load(open(os.path.join(GOLD, "scoring_golden.json")))
    for i in inp["ids"]:
        i = str(i)
        segs = [inp["seg_%s_%d" % (i, a)] for a in range(int(inp["nseg_" + i]))]
        s = boundary_scores(seg(inp["img_" + i]), segs)
        ref = scores[i + "/oracle"]
        assert s["recall"] == ref["recall"] and s["precision"] == ref["precision"]
        assert s["fmeasure"] == 2 * ref["precision"] * ref["recall"] / (ref["precision"] + ref["recall"])


def test_full_size_batch_vs_c_oracle_and_properties(seg):
    """Batch of 6 synthetic 481x321 images: two checked pixel-for-pixel against the C oracle, all
    against batch/single consistency, label range and determinism."""
    from gabor_color_image_segmentation_amd.synthetic import synthetic_batch
    imgs = synthetic_batch(6, 321, 481, seed=0)
    out = seg.segment_batch(imgs)
    assert out.shape == (6, 321, 481) and out.min() >= 0 and out.max() <= 7
    ref = co.segment_batch(imgs[[0, 5]], seg.bank.tapq, seg.bank.shift, 6)
    assert np.array_equal(out[0], ref[0]) and np.array_equal(out[5], ref[1])
    assert np.array_equal(seg(imgs[3]), out[3])                      # segment == segment_batch row
    assert np.array_equal(seg.segment_batch(imgs), out)              # deterministic
    assert np.array_equal(seg.segment_batch(imgs[::-1])[::-1], out)  # per-image: order independent


def test_global_codebook_full_size_vs_c_oracle(seg):
    from gabor_color_image_segmentation_amd.synthetic import synthetic_batch
    imgs = synthetic_batch(3, 321, 481, seed=9)
    got = seg.segment_batch(imgs, mode="global")
    assert np.array_equal(got, co.segment_batch(imgs, seg.bank.tapq, seg.bank.shift, 6, mode="global"))


def test_k16_two_mfma_tiles_and_portrait(built):
    from gabor_color_image_segmentation_amd import Segmenter
    from gabor_color_image_segmentation_amd.synthetic import synthetic_batch
    imgs = synthetic_batch(2, 481, 321, seed=4)
    s = Segmenter(k=16, n_iter=4)
    got = s.segment_batch(imgs)
    ref = co.segment_batch(imgs, s.bank.tapq, s.bank.shift, 6, k=16, n_iter=4)
    assert np.array_equal(got, ref)


def test_wide_feature_vectors_use_the_generic_pass(built):
    """D = 210 >= 208 planes: the non-MFMA k-means pass; still bit-exact."""
    from gabor_color_image_segmentation_amd import Segmenter
    from gabor_color_image_segmentation_amd.synthetic import synthetic_batch
    imgs = synthetic_batch(2, 40, 56, seed=6)
    s = Segmenter(n_scales=7, n_orient=10, k=4, n_iter=3)
    assert s.bank.n_features == 210
    got = s.segment_batch(imgs)
    for b in range(2):
        assert np.array_equal(got[b], so.segment(imgs[b], n_scales=7, n_orient=10, k=4, n_iter=3))


def test_sixty_four_filter_bank_features(built):
    """BASELINE config 4's 8x8 bank (F = 64, D = 192): three MFMA row-tile launches."""
    import torch
    from gabor_color_image_segmentation_amd import Segmenter
    from gabor_color_image_segmentation_amd.synthetic import synthetic_batch
    imgs = synthetic_batch(1, 48, 72, seed=12)
    s = Segmenter(n_scales=8, n_orient=8)
    got = s.features_device(torch.from_numpy(imgs).cuda()).cpu().numpy().view(np.uint16)[0]
    tapq, shift = so.bank(8, 8)
    assert np.array_equal(got, co.gabor_features(imgs[0], tapq, shift, 8))


def test_degenerate_inputs(seg):
    """Constant image: every feature equal, all pixels tie -> label 0 everywhere (lowest index)."""
    img = np.full((32, 48, 3), 77, np.uint8)
    assert np.array_equal(seg(img), np.zeros((32, 48), np.int32))
    with pytest.raises(ValueError):
        seg(np.zeros((7, 40, 3), np.uint8))


def test_integration_md_ctypes_stub_runs_and_matches_the_oracle(built):
    """The binding shown to a maintainer in INTEGRATION.md is executed verbatim."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "INTEGRATION.md")).read()
    code = re.search(r"```python\n(import ctypes as C.*?)```", text, re.S).group(1)
    ns = {}
    cwd = os.getcwd()
    os.chdir(root)                       # the stub loads the library by its repo-relative path
    try:
        exec(compile(code, "INTEGRATION.md", "exec"), ns)
        from gabor_color_image_segmentation_amd.synthetic import synthetic_batch
        img = synthetic_batch(1, 56, 88, seed=17)[0]
        got = ns["segment"](img, k=6, n_iter=4)
    finally:
        os.chdir(cwd)
    assert np.array_equal(got, so.segment(img, k=6, n_iter=4))


def test_sixty_four_filter_bank_full_segment(built):
    """BASELINE config 4 end to end (D = 192: four Gabor launches, wide MFMA k-means pass)."""
    from gabor_color_image_segmentation_amd import Segmenter
    from gabor_color_image_segmentation_amd.synthetic import synthetic_batch
    imgs = synthetic_batch(2, 40, 64, seed=23)
    s = Segmenter(n_scales=8, n_orient=8, k=6, n_iter=3)
    got = s.segment_batch(imgs, mode="global")
    tapq, shift = so.bank(8, 8)
    assert np.array_equal(got, co.segment_batch(imgs, tapq, shift, 8, k=6, n_iter=3, mode="global"))


def test_batch64_full_size_global_codebook_vs_c_oracle(seg):
    """The configuration bench.py times (BASELINE config 2: 64 synthetic 481x321 images, seed 0, 4x6 bank, k = 8,
    n_iter = 10, one global codebook; 768 k-means workgroups with 12 partial rows per image, reverse sweeps): EVERY
    pixel of all 64 label maps against the C oracle (OpenMP over the host cores, ~20 s on the GPU box's 16)."""
    from gabor_color_image_segmentation_amd.synthetic import synthetic_batch
    imgs = synthetic_batch(64, 321, 481, seed=0)
    got = seg.segment_batch(imgs, mode="global")
    ref = co.segment_batch(imgs, seg.bank.tapq, seg.bank.shift, 6, mode="global")
    assert got.shape == ref.shape == (64, 321, 481)
    assert np.array_equal(got, ref), f"{(got != ref).mean():.4%} of pixels differ"
    assert len(np.unique(got)) == 8


def test_batch64_full_size_per_image_codebooks_vs_c_oracle(seg):
    """Same batch, the reference's semantics (script.py:22-38: every image on its own): six of the 64 label maps
    (first, last and four in between) against the C oracle."""
    from gabor_color_image_segmentation_amd.synthetic import synthetic_batch
    imgs = synthetic_batch(64, 321, 481, seed=0)
    got = seg.segment_batch(imgs)
    pick = [0, 7, 21, 42, 50, 63]
    ref = co.segment_batch(imgs[pick], seg.bank.tapq, seg.bank.shift, 6)
    assert np.array_equal(got[pick], ref)


def test_config4_full_size_vs_c_oracle(built):
    """BASELINE config 4 (8x8 = 64-filter bank, D = 192, four pyramid levels) at full image size: features of one
    481x321 image and the global-codebook labels of a batch of two, against the C oracle."""
    import torch
    from gabor_color_image_segmentation_amd import Segmenter
    from gabor_color_image_segmentation_amd.synthetic import synthetic_batch
    imgs = synthetic_batch(2, 321, 481, seed=44)
    s = Segmenter(n_scales=8, n_orient=8, n_iter=5)
    tapq, shift = so.bank(8, 8)
    got = s.features_device(torch.from_numpy(imgs[:1]).cuda()).cpu().numpy().view(np.uint16)[0]
    assert np.array_equal(got, co.gabor_features(imgs[0], tapq, shift, 8))
    lab = s.segment_batch(imgs, mode="global")
    assert np.array_equal(lab, co.segment_batch(imgs, tapq, shift, 8, n_iter=5, mode="global"))


def test_bsd_val_boundary_f_gate(seg):
    """The quality gate: 24 decoded BSD500 val images through the HIP path give the oracle's label
    maps bit for bit, and the batched GPU scorer (gcs_boundary_counts_batch / gcs_region_counts_batch against the packed
    ground truth of all 500 ids) gives exactly the P / R / F the reference's own metrics class
    (/root/reference/BSD_metrics/metrics.py:58-96, run by tests/golden/make_bsd_val_scores.py) recorded for them."""
    import torch
    from gabor_color_image_segmentation_amd.evaluate_gpu import all_scores_batch_device
    from gabor_color_image_segmentation_amd.groundtruth import PackedTruth
    doc = json.load(open(os.path.join(GOLD, "bsd_val_scores.json")))
    pack = np.load(os.path.join(GOLD, "bsd_val_images.npz"))
    pt = PackedTruth(os.path.join(GOLD, "bsd500_truth.npz"))
    ids = [str(i) for i in pack["ids"]]
    f_gpu = {}
    for shape in ((321, 481), (481, 321)):
        group = [i for i in ids if pack["img_" + i].shape[:2] == shape]
        assert group
        imgs = torch.from_numpy(np.stack([pack["img_" + i] for i in group])).cuda()
        labs = seg.segment_device(imgs)                                  # per-image codebooks: script.py:22-30
        host = labs.cpu().numpy()
        for b, i in enumerate(group):
            assert np.array_equal(host[b], pack["labels_" + i]), i
        scored = all_scores_batch_device(labs, *pt.stack(group))
        # resident ground truth (prepared once per group, nothing uploaded per call), n_segments = the plan's k: the same floats
        assert all_scores_batch_device(labs, pt.to_device(group), n_segments=seg.k) == scored
        for i, g in zip(group, scored):
            ref = doc["per_id"][i]["v2"]
            assert g["regions"] == ref["regions"]
            for key in ("recall", "precision", "fmeasure", "density"):
                assert g[key] == ref[key], (i, key, g[key], ref[key])
            for key in ("underseg", "undersegNP", "compactness"):
                assert abs(g[key] - ref[key]) <= 1e-12, (i, key)
            f_gpu[i] = g["fmeasure"]
    assert float(np.mean([f_gpu[i] for i in ids])) == float(np.mean([doc["per_id"][i]["v2"]["fmeasure"] for i in ids]))
