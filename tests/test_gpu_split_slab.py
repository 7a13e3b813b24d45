"""GPU parity of the SPLIT feature slab (round 6; csrc/common.h): values of 4096 and more live in the TOP array and are read by a
Lloyd pass only where the tile's flag word says so. These cases put such values where the format can go wrong: every tile flagged
(a full-contrast grating), a handful of flagged tiles among clean ones, flagged tiles inside packed edge strips, flagged tiles on
some ranks only (their sums must reach the all-reduced centroids) - and they read the flag words themselves: set exactly where
the canonical features hold a value >= 4096 (no false negative: exactness; no false positive: the bytes saved)."""
import os

import numpy as np
import pytest

from oracle import spec_oracle as so

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda(built):
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch


def _synth(b, h, w, seed):
    from gabor_color_image_segmentation_amd.synthetic import synthetic_batch
    return synthetic_batch(b, h, w, seed=seed)


def _grating(h, w, f=0.4, lo=0, hi=255, vertical=True):
    yy, xx = np.mgrid[0:h, 0:w]
    g = np.where(np.sin(2 * np.pi * f * (xx if vertical else yy)) >= 0, hi, lo).astype(np.uint8)
    return np.stack([g, g, g], -1)


def _with_patch(img, y0, x0, size=16):
    """A small full-contrast grating patch at the bank's peak frequency: a few values >= 4096 under it, none elsewhere."""
    out = img.copy()
    y0, x0 = min(y0, img.shape[0] - size), min(x0, img.shape[1] - size)
    out[y0:y0 + size, x0:x0 + size] = _grating(size, size)
    return out


def _tile_of_pixels(h, w):
    """(H, W) int array: the slab tile of every pixel for banks of at most two levels (csrc/common.h: main 8x8 blocks in raster
    order, then the virtual blocks of a packed right edge of 1 - 2 columns and a packed bottom edge of 1 - 2 rows)."""
    pack_r = h >= 8 and w >= 8 and (w & 7) in (1, 2)
    pack_b = h >= 8 and w >= 8 and (h & 7) in (1, 2)
    bx_n = w // 8 if pack_r else (w + 7) // 8
    by_n = h // 8 if pack_b else (h + 7) // 8
    wm = 8 * bx_n if pack_r else 1 << 29
    hm = 8 * by_n if pack_b else 1 << 29
    n_r = ((h + 1) // 2 + 15) // 16 if pack_r else 0
    nmain = bx_n * by_n
    y, x = np.mgrid[0:h, 0:w]
    blk = (y >> 3) * bx_n + (x >> 3)
    blk = np.where(x >= wm, nmain + ((y >> 1) >> 4), np.where(y >= hm, nmain + n_r + ((x >> 1) >> 4), blk))
    return blk >> 2


def _flags(seg, feats, b, h, w):
    """The flag words of a split slab -> bool [b][ntiles] (tests may know the layout: tile_bytes = 2 S, flags behind 2 S ntiles)."""
    lib = seg.ops.lib
    ns, no = seg.bank.n_scales, seg.bank.n_orient
    img_bytes = lib.gcs_feature_slab_bytes(1, h, w, ns, no)
    assert lib.gcs_feature_pass_bytes(1, h, w, ns, no) * 4 < img_bytes * 3 + 4, "this bank does not take the split slab"
    levels = [(3 * min(2, ns - 2 * L) * no, 256 >> (2 * L)) for L in range((ns + 1) // 2)]
    s = sum(d * n for d, n in levels)
    ntiles = img_bytes // (2 * s)
    raw = feats.cpu().numpy().view(np.uint8)[:b * img_bytes].reshape(b, img_bytes)
    return raw[:, 2 * s * ntiles:2 * s * ntiles + 4 * ntiles].reshape(b, ntiles, 4).any(axis=2), ntiles


@pytest.mark.parametrize("h,w", [(64, 96), (81, 121), (137, 82), (321, 481)])
def test_flag_words_are_set_exactly_where_a_value_needs_its_top_nibble(torch_cuda, h, w):
    """Clean images, a grating (every tile flagged) and images with one small full-contrast patch - in the main blocks, and, for the
    shapes with packed edge strips, on the right and bottom edge -: features == oracle, and a tile's flag is set if and only if
    one of its values is 4096 or more."""
    from gabor_color_image_segmentation_amd import Segmenter
    torch = torch_cuda
    base = _synth(3, h, w, seed=201)
    imgs = np.stack([base[0], _grating(h, w), _with_patch(base[1], h // 3, w // 4), _with_patch(base[2], h - 16, w - 16),
                     _with_patch(base[0], h - 16, 3), _with_patch(base[1], 5, w - 16)])
    seg = Segmenter()
    b = len(imgs)
    feats = seg.ops.feature_slab(b, h, w)
    seg.ops.gabor_features(torch.from_numpy(imgs).cuda(), feats)
    got = seg.ops.features_unpack(feats, b, h, w).cpu().numpy().view(np.uint16)
    tapq, shift = so.bank()
    flags, ntiles = _flags(seg, feats, b, h, w)
    tile = _tile_of_pixels(h, w)
    assert tile.max() + 1 == ntiles
    n_flagged = []
    for i in range(b):
        ref = so.gabor_features(imgs[i], tapq, shift, 6)
        assert np.array_equal(got[i], ref), i
        want = np.zeros(ntiles, bool)
        want[np.unique(tile[(ref >= 4096).any(axis=0)])] = True
        assert np.array_equal(flags[i], want), (i, int(flags[i].sum()), int(want.sum()))
        n_flagged.append(int(want.sum()))
    assert n_flagged[1] == ntiles and all(0 < n < ntiles // 2 for n in n_flagged[2:]), n_flagged   # the cases are what they claim


@pytest.mark.parametrize("mode", ["per_image", "global"])
def test_labels_with_flagged_tiles_equal_the_c_oracle(torch_cuda, mode):
    """A batch that mixes clean images, gratings (every value path through TOP) and patched images, both codebook modes, forward
    and reverse sweeps (6 passes): every label == the C oracle. The centroids of the global run come to lie above 4096 in some
    planes (the gratings), i.e. the assign patterns' high digits leave the 4-bit range too."""
    from oracle import c_oracle as co
    from gabor_color_image_segmentation_amd import Segmenter
    h, w = 137, 201
    base = _synth(4, h, w, seed=77)
    imgs = np.stack([_grating(h, w), base[0], _with_patch(base[1], 40, 60), _grating(h, w, f=0.2828, vertical=False), base[2],
                     _with_patch(base[3], h - 16, w - 16), _grating(h, w, lo=40, hi=230)])
    seg = Segmenter(n_iter=6)
    got = seg.segment_batch(imgs, mode=mode)
    ref = co.segment_batch(imgs, seg.bank.tapq, seg.bank.shift, seg.bank.n_orient, k=8, n_iter=6, mode=mode)
    assert np.array_equal(got, ref) and len(np.unique(ref)) > 1


def _rank_worker(rank, world, port, shard_file, out_dir):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from gabor_color_image_segmentation_amd import Segmenter
        imgs = np.load(shard_file)[f"r{rank}"]
        seg = Segmenter(n_iter=5, device="cuda:0")
        lab = seg.segment_device(torch.from_numpy(imgs).cuda(), mode="global").cpu().numpy()
        np.save(os.path.join(out_dir, f"lab_{rank}.npy"), lab)
    finally:
        dist.destroy_process_group()


def test_flagged_tiles_on_some_ranks_reach_the_all_reduced_sums(tmp_path, built):
    """Three ranks share cuda:0 (gloo carries the int64 all-reduce): rank 0 holds the image the init centroids come from, rank 1
    the gratings and a patched image (flagged tiles), rank 2 clean images. The global-codebook labels of all ranks == the C
    oracle on the unsharded batch - the sums of the flagged tiles took part in every update."""
    import torch.multiprocessing as mp
    from oracle import c_oracle as co
    h, w = 96, 136
    base = _synth(5, h, w, seed=311)
    shards = [np.stack([base[0], base[1]]), np.stack([_grating(h, w), _with_patch(base[2], 30, 40)]), np.stack([base[3], base[4]])]
    shard_file = str(tmp_path / "shards.npz")
    np.savez(shard_file, **{f"r{r}": s for r, s in enumerate(shards)})
    port = 40500 + (os.getpid() % 2000)
    mp.spawn(_rank_worker, args=(3, port, shard_file, str(tmp_path)), nprocs=3, join=True)
    got = np.concatenate([np.load(tmp_path / f"lab_{r}.npy") for r in range(3)])
    tapq, shift = so.bank()
    ref = co.segment_batch(np.concatenate(shards), tapq, shift, 6, k=8, n_iter=5, mode="global")
    assert np.array_equal(got, ref)
