#!/usr/bin/env python3
"""Phase timeline of a Lloyd pass from in-kernel wall_clock64 stamps (100 MHz); optional 4th argument: batch size (1 = the pass of a
one-image call). The stamps are NOT in the product: apply
tools/dbg/patches/pass_stamps.patch, build a variant (tools/build_variant.sh stamps), run this on it, drop the patch.
    git apply tools/dbg/patches/pass_stamps.patch && bash tools/build_variant.sh stamps && git checkout gabor_color_image_segmentation_amd/csrc/kmeans.hip
    python tools/dbg/nv_stamps.py build_ab/stamps.so [n_scales n_orient]
Slots per workgroup: 0 entry, 1 first tile's loads issued (deep-bank pass), 2 centroid gather done, 3 barrier, 4 key bases done,
5 A fragments done, 6 prologue done (barrier), 8 tile loop done, 9 fold done, 15 HW_REG_XCC_ID."""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gabor_color_image_segmentation_amd import _lib, Segmenter
from gabor_color_image_segmentation_amd.synthetic import synthetic_shard
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
ns, no = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (8, 8)
B = int(sys.argv[4]) if len(sys.argv) > 4 else 64              # (B = 1: the pass of a one-image call)
H, W = 321, 481
imgs = torch.from_numpy(synthetic_shard(0, B, H, W)).cuda()
seg = Segmenter(n_scales=ns, n_orient=no)
for _ in range(3):
    seg.segment_device(imgs, mode="global")
ws = seg._workspace(B, H, W, "global")
lib = ctypes.CDLL(_lib.LIB_PATH)
for rep in range(3):
    seg.ops.assign_accumulate(ws["feats"], ws["cent"], B, H, W, seg.k, 1, None, ws["partials"], reverse=bool(rep & 1))
    torch.cuda.current_stream().synchronize()
    buf = np.zeros(1024 * 16, dtype=np.uint64)
    assert lib.gcs_debug_nv_stamps(buf.ctypes.data_as(ctypes.c_void_p)) == 0
    nwg = int((buf.reshape(1024, 16)[:, 0] > 0).sum()) if B < 64 else 768
    st = buf.reshape(1024, 16)[:nwg].astype(np.int64)
    xcc = st[:, 15] & 15
    us = (st[:, :10] - st[:, 0].min()) / 100.0
    rel = (st[:, :10] - st[:, 0:1]) / 100.0
    loop = us[:, 8] - us[:, 6]
    print(f"pass {rep} ({'reverse' if rep & 1 else 'forward'}), bank {ns}x{no}: entries within {us[:,0].max():.1f} us; prologue {np.median(rel[:,6]):.1f}; "
          f"tile loop done min {us[:,8].min():.1f} p10 {np.percentile(us[:,8],10):.1f} median {np.median(us[:,8]):.1f} p90 {np.percentile(us[:,8],90):.1f} "
          f"max {us[:,8].max():.1f}; fold {np.median(us[:,9]-us[:,8]):.1f}; kernel end {us[:,9].max():.1f} us")
    if st[:, 2].max() == 0 and st[:, 3].max() > 0:               # the 4x6 pass: no stamp 1 / 2 (its first loads go out after the prologue)
        names = {3: "centroid gather done + barrier", 4: "key bases done", 5: "A fragments done", 6: "barrier"}
        print("   prologue, median us after entry: " + ", ".join(f"{n} {np.median(rel[:, i]):.2f}" for i, n in names.items()))
    if st[:, 2].max() > 0:
        names = {1: "first loads issued", 2: "centroid gather done", 3: "barrier", 4: "key bases done", 5: "A fragments done", 6: "barrier"}
        print("   prologue, median us after entry: " + ", ".join(f"{n} {np.median(rel[:, i]):.2f}" for i, n in names.items()))
    print("   tile loop by XCD (median / max us): " + "  ".join(f"x{x}: {np.median(loop[xcc==x]):.0f}/{loop[xcc==x].max():.0f}" for x in range(8)))
    if nwg == 768:
        gen = [loop[:256], loop[256:512], loop[512:]]
        print("   tile loop by dispatch generation (workgroups 0-255 / 256-511 / 512-767), median us: " + " / ".join(f"{np.median(x):.0f}" for x in gen))
    else:
        print(f"   {nwg} workgroups; tile loop median {np.median(loop):.2f} us (min {loop.min():.2f}, max {loop.max():.2f})")
