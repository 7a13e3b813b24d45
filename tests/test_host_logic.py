"""Host orchestration (grouping, Lloyd schedule, argument checks) with the CPU fake backend."""
import numpy as np
import pytest
import torch

from gabor_color_image_segmentation_amd import Segmenter, make_bank
from gabor_color_image_segmentation_amd.synthetic import synthetic_batch, synthetic_shard
from oracle import spec_oracle as so
from fake_ops import OracleOps


def _seg(**kw):
    bank_kw = {a: kw[a] for a in ("n_scales", "n_orient", "ksize") if a in kw}
    return Segmenter(ops=OracleOps(make_bank(**bank_kw)), **kw)


def test_per_image_groups_equal_the_oracle():
    imgs = synthetic_batch(5, 24, 40, seed=1)
    seg = _seg(n_iter=3)
    out = seg.segment_device(torch.from_numpy(imgs), mode="per_image", group=2).numpy()
    assert [c for c in seg.ops.calls if c[0] == "gabor"] == [("gabor", 2), ("gabor", 2), ("gabor", 1)]
    for b in range(5):
        assert np.array_equal(out[b], so.segment(imgs[b], n_iter=3))


def test_global_mode_equals_the_oracle():
    imgs = synthetic_batch(3, 24, 40, seed=2)
    out = _seg(n_iter=4, k=5).segment_device(torch.from_numpy(imgs), mode="global").numpy()
    assert np.array_equal(out, so.segment_batch(imgs, mode="global", k=5, n_iter=4))


def test_n_iter_one_returns_the_init_assignment():
    imgs = synthetic_batch(1, 16, 24, seed=3)
    out = _seg(n_iter=1).segment_device(torch.from_numpy(imgs)).numpy()
    assert np.array_equal(out[0], so.segment(imgs[0], n_iter=1))


def test_argument_checks():
    seg = _seg()
    with pytest.raises(ValueError):
        seg.segment_device(torch.zeros((1, 16, 16, 3), dtype=torch.float32))
    with pytest.raises(ValueError):
        seg.segment_device(torch.zeros((16, 16, 3), dtype=torch.uint8))
    with pytest.raises(ValueError):
        seg.segment_device(torch.zeros((1, 7, 16, 3), dtype=torch.uint8))
    with pytest.raises(ValueError):
        seg.segment_device(torch.zeros((1, 16, 16, 3), dtype=torch.uint8), mode="batch")
    with pytest.raises(ValueError):
        seg(np.zeros((16, 16), np.uint8))
    with pytest.raises(ValueError):
        seg(np.zeros((16, 16, 3), np.float32))
    with pytest.raises(ValueError):
        Segmenter(k=17, ops=object())
    with pytest.raises(ValueError):
        Segmenter(n_iter=0, ops=object())


def test_synthetic_shards_tile_the_global_batch():
    full = synthetic_batch(5, 16, 24, seed=7)
    assert np.array_equal(np.concatenate([synthetic_shard(0, 2, 16, 24, seed=7), synthetic_shard(2, 3, 16, 24, seed=7)]), full)
    assert full.dtype == np.uint8 and len({im.tobytes() for im in full}) == 5


def test_connectivity_option_relabels_connected_regions():
    imgs = synthetic_batch(2, 24, 40, seed=9)
    seg = Segmenter(ops=OracleOps(make_bank()), n_iter=3, connectivity=True)
    out = seg.segment_device(torch.from_numpy(imgs)).numpy()
    for b in range(2):
        assert np.array_equal(out[b], so.connected_regions(so.segment(imgs[b], n_iter=3)))


def test_segment_images_groups_by_shape_and_keeps_the_input_order():
    """Segmenter.segment_images (the data-set form of the loop at script.py:22-38): mixed shapes are batched per shape, every
    label map equals the one-image call's, results come out in input order, the batches are as large as asked for."""
    shapes = [(16, 24), (24, 16), (16, 24), (16, 24), (24, 16), (17, 24), (16, 24)]
    imgs = [synthetic_batch(1, h, w, seed=20 + i)[0] for i, (h, w) in enumerate(shapes)]
    seg = _seg(n_iter=2)
    got = list(seg.segment_images(iter(imgs), batch=3))
    assert len(got) == len(imgs)
    for g, im in zip(got, imgs):
        assert g.shape == im.shape[:2] and np.array_equal(g, so.segment(im, n_iter=2))
    # 4 images of 16x24 -> one batch of 3 and a remainder of 1; 2 of 24x16; 1 of 17x24
    assert sorted(c[1] for c in seg.ops.calls if c[0] == "gabor") == [1, 1, 2, 3]
    with pytest.raises(ValueError):
        list(seg.segment_images([np.zeros((16, 24), np.uint8)]))
    with pytest.raises(ValueError):
        list(seg.segment_images(imgs, batch=0))


def test_segment_images_bounds_what_waits_behind_a_rare_shape():
    """image 0 is the only one of its shape. Without a bound nothing would be yielded before the input ends and
    every later label map would pile up; with it the rare shape's partial group is flushed once 2 * batch finished maps
    wait, and the output resumes while the loader is still running."""
    batch, n = 2, 30
    rare = synthetic_batch(1, 24, 16, seed=40)[0]
    common = [synthetic_batch(1, 16, 24, seed=41 + i)[0] for i in range(n)]
    seg = _seg(n_iter=2)
    taken, yielded_at = [0], []

    def loader():
        yield rare
        for im in common:
            taken[0] += 1
            yield im

    got = []
    for lab in seg.segment_images(loader(), batch=batch):
        yielded_at.append(taken[0])
        got.append(lab)
    assert len(got) == n + 1
    assert np.array_equal(got[0], so.segment(rare, n_iter=2))
    for g, im in zip(got[1:], common):
        assert np.array_equal(g, so.segment(im, n_iter=2))
    # the first result left while at most 2 * batch + batch images had been taken from the loader, not at its end
    assert yielded_at[0] <= 3 * batch + 1 < n


def test_debug_switches_parse_and_reject_unknown_names():
    """The one set of measurement / test switches (GCS_DEBUG=...): every name is documented in segmenter.DebugSwitches, an
    unknown one is an error instead of a silently ignored typo, and a plan carries its own copy."""
    from gabor_color_image_segmentation_amd.segmenter import DebugSwitches
    d = DebugSwitches("no_reverse, slab_candidates=3,force_collectives")
    assert d.no_reverse and d.force_collectives and not d.no_graph and d.slab_candidates == 3
    assert not any(vars(DebugSwitches("")).values())
    with pytest.raises(ValueError):
        DebugSwitches("no_revrse")
    with pytest.raises(ValueError):
        DebugSwitches("no_graph=1")
    a, b = _seg(), _seg()
    a.debug.force_collectives = True
    assert not b.debug.force_collectives


def test_lds_bank_model_matches_the_deep_bank_kernel_constants():
    """tools/design/lds_bank_model.py restates the LDS accesses of kmeans_pass_native_kernel against the lane groups the LDS serves
    (profiles/r5_notes.md). It must model the layout that ships: its row pitches and A-table slot count are the kernel's, every read
    pattern is conflict-free in the model, and the first round-5 build (--first) reproduces its 624 extra cycles per tile."""
    import os
    import re
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, "gabor_color_image_segmentation_amd", "csrc", "kmeans.hip")).read()
    m = re.search(r"NV_P1 = (\d+) \+ (\d+), NV_P2 = (\d+) \+ (\d+), NV_P3 = (\d+);", src)
    p1, p2 = int(m.group(1)) + int(m.group(2)), int(m.group(3)) + int(m.group(4))
    nsl = int(re.search(r"constexpr int NV_APAT_SLOTS = (\d+);", src).group(1))
    model = open(os.path.join(root, "tools", "design", "lds_bank_model.py")).read()
    assert f"else ({p1}, {p2}, {nsl})" in model, (p1, p2, nsl)
    tool = os.path.join(root, "tools", "design", "lds_bank_model.py")
    out = subprocess.run([sys.executable, tool], capture_output=True, text=True, check=True).stdout
    rows = {l[:42].strip(): int(l[42:].split()[0]) for l in out.splitlines() if "extra LDS cycles" in l}
    assert all(v == 0 for k, v in rows.items() if not k.startswith("staging")), rows
    assert rows["staging  ds_write_b128 / 2 x b64"] <= 30
    first = subprocess.run([sys.executable, tool, "--first"], capture_output=True, text=True, check=True).stdout
    assert re.search(r"total\s+624", first), first


def test_lds_bank_model_of_the_split_slab_pass_matches_its_constants_and_the_measured_conflicts():
    """tools/design/lds_bank_model_narrow.py restates the LDS accesses of the split-slab Lloyd pass with level 1 kept compact
    (round 6) against the lane groups the LDS serves. Its pitches and swizzle are the kernel's; the transposed reads, the 16-byte
    update reads and the label reads are conflict-free; and the three builds of round 6 reproduce what SQ_LDS_BANK_CONFLICT measured
    per tile (338 / 223 / 165 cycles: profiles/r6_notes.md) within 5 %."""
    import os
    import re
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, "gabor_color_image_segmentation_amd", "csrc", "kmeans.hip")).read()
    model = open(os.path.join(root, "tools", "design", "lds_bank_model_narrow.py")).read()
    assert "constexpr int KP_PITCH = KP_TP * 2 + 64;" in src and "KP_PITCH = KP_TP * 2 + 64" in model
    assert "constexpr int KP_P1 = 128 + 48;" in src and "KP_P1 = 128 + 48" in model
    assert "((r >> 3) & 1) * 32" in src and "((r >> 3) & 1) * 32" in model                  # row_swz
    assert "GCS_KP_LAUNCHS(1, 3, 2)" in src and "L0T = 0 if B64 else 2" in model            # 16-byte reads for two plane tiles
    tool = os.path.join(root, "tools", "design", "lds_bank_model_narrow.py")

    def run(*flags):
        out = subprocess.run([sys.executable, tool, *flags], capture_output=True, text=True, check=True).stdout
        rows = {l[:48].strip(): int(l[48:].split()[0]) for l in out.splitlines() if "extra LDS cycles" in l or l.startswith("total")}
        return rows

    ships = run()
    assert ships["assign  ds_read_b64_tr_b16"] == 0 and ships["update  ds_read_b128 (plane tiles of level 0)"] == 0
    assert ships["update  labels ds_read_b64"] == 0 and ships["update  2 x ds_read_b64"] <= 16
    for flags, measured in (((), 165), (("--b64",), 223), (("--b64", "--noswz"), 338)):
        total = run(*flags)["total"]
        assert abs(total - measured) <= 0.05 * measured, (flags, total, measured)
