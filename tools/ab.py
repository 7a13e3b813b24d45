#!/usr/bin/env python3
"""Same-box A/B of libgcs.so builds (cdna guide §5.4 rule 24: interleaved rounds in ONE process).

    python tools/ab.py name=path/to/libgcs.so [name=...] [--rounds 15] [--batch 64] [--bank 4x6]

Every build gets its own slab (layouts may differ between builds), runs its own Gabor stage, and is timed with HIP
events, the builds taking turns inside each round: the Gabor stage, one forward and one reverse Lloyd pass, and the whole
10-pass step. Prints median / min per build. Results are checked for equality across builds (labels of the step)."""
import argparse, os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gabor_color_image_segmentation_amd import _lib, Segmenter
from gabor_color_image_segmentation_amd.synthetic import synthetic_shard

ap = argparse.ArgumentParser()
ap.add_argument("libs", nargs="+")
ap.add_argument("--rounds", type=int, default=15)
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--bank", default="4x6")
ap.add_argument("--mode", default="global")
ap.add_argument("--size", default="321x481")
args = ap.parse_args()
ns, no = map(int, args.bank.split("x"))
H, W = map(int, args.size.split("x"))
B = args.batch
imgs = torch.from_numpy(synthetic_shard(0, B, H, W)).cuda()
segs = {}
import ctypes
for spec in args.libs:
    name, path = spec.split("=", 1)
    _lib._lib = None
    _lib.LIB_PATH = os.path.abspath(path)
    # an older build may be compared as long as the entry points used here kept their signatures (ABI 15 -> 16 only dropped
    # gcs_labels_raster_u8 and turned the label slab into a raster map)
    raw = ctypes.CDLL(_lib.LIB_PATH)
    _lib.ABI_VERSION = raw.gcs_abi_version()
    if not hasattr(_lib, "_ALL_SIGNATURES"):
        _lib._ALL_SIGNATURES = dict(_lib.SIGNATURES)
    _lib.SIGNATURES = {k: v for k, v in _lib._ALL_SIGNATURES.items() if hasattr(raw, k)}   # an older build lacks the newest entry points
    segs[name] = Segmenter(n_scales=ns, n_orient=no)
n_sets = B if args.mode == "per_image" else 1


def ev():
    return torch.cuda.Event(enable_timing=True)


res = {n: dict(gabor=[], fwd=[], rev=[], step=[]) for n in segs}
labels = {}
for rnd in range(args.rounds + 2):
    for name, seg in segs.items():
        ws = seg._workspace(B, H, W, args.mode)
        k = seg.k
        e = [ev() for _ in range(8)]
        e[0].record(); seg.ops.gabor_features(imgs, ws["feats"]); e[1].record()
        seg.ops.kmeans_init(ws["feats"], B, H, W, k, n_sets, ws["cent"])
        # sums only, as every pass of a step but the last (labels=None): the label output differs between ABI versions
        e[2].record(); seg.ops.assign_accumulate(ws["feats"], ws["cent"], B, H, W, k, n_sets, None, ws["partials"]); e[3].record()
        e[4].record(); seg.ops.assign_accumulate(ws["feats"], ws["cent"], B, H, W, k, n_sets, None, ws["partials"], reverse=True); e[5].record()
        e[6].record(); out = seg.segment_device(imgs, mode=args.mode); e[7].record()
        torch.cuda.synchronize()
        if rnd >= 2:
            r = res[name]
            r["gabor"].append(e[0].elapsed_time(e[1])); r["fwd"].append(e[2].elapsed_time(e[3]))
            r["rev"].append(e[4].elapsed_time(e[5])); r["step"].append(e[6].elapsed_time(e[7]))
        labels[name] = out
names = list(segs)
for n in names[1:]:
    print("labels equal to", names[0], ":", n, bool(torch.equal(labels[n], labels[names[0]])))
for name, r in res.items():
    print(f"{name:12s} " + "  ".join(f"{k} med {statistics.median(v):.4f} min {min(v):.4f}" for k, v in r.items()) +
          f"  | {B * H * W / statistics.median(r['step']) / 1e3:.0f} Mpix/s")
