"""Kernel timeline of the Gabor stage, B images of H x W (default 64 x 321 x 481). Run under
`rocprofv3 --kernel-trace --output-format csv -d DIR -- python tools/gabor_stage_probe.py [steps] [B H W]` and show the timeline of the
last stage with `python tools/gabor_stage_probe.py show DIR`.

    steps   STEADY STATE (round 5): whole segment_device steps back to back, as bench.py times them - the
            stage starts behind the previous step's last Lloyd pass, its side stream is warm, the chip's clocks are where a
            running job holds them. Prints the stage as HIP events inside those steps measure it.
    (none)  the stage alone, 12 isolated calls with a synchronisation after each (round 4's probe: a cold, idle chip in front
            of every call - 0.51 ms where the steady state measures 0.40 - 0.42)."""
import os, sys
if len(sys.argv) > 2 and sys.argv[1] == "show":
    import csv, glob
    rows = []
    for f in glob.glob(sys.argv[2] + "/**/*kernel_trace.csv", recursive=True):
        rows += list(csv.DictReader(open(f)))
    ker = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60], r.get("Grid_Size", "")) for r in rows)
    ker = [k for k in ker if "gabor" in k[2]]
    idx = [i for i, k in enumerate(ker) if "gabor_plane" in k[2]]
    i0 = idx[-1]
    t0 = ker[i0][0]
    for s, e, n, g in ker[i0:]:
        print("%8.1f %8.1f (%6.1f us) grid %8s  %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, g, n))
    print("stage span %.1f us" % ((max(k[1] for k in ker[i0:]) - t0) / 1e3))
    sys.exit(0)
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gabor_color_image_segmentation_amd import Segmenter
from gabor_color_image_segmentation_amd.synthetic import synthetic_shard
argv = sys.argv[1:]
steady = bool(argv) and argv[0] == "steps"
if steady:
    argv = argv[1:]
B, H, W = (int(x) for x in (argv[:3] + ["64", "321", "481"][len(argv):]))
if steady:
    seg = Segmenter()
    imgs = torch.from_numpy(synthetic_shard(0, B, H, W)).cuda()
    ops, ev = seg.ops, []

    class Timed:
        def __getattr__(self, n):
            return getattr(ops, n)

        def gabor_features(self, *a, **kw):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record(); ops.gabor_features(*a, **kw); e.record()
            ev.append((s, e))
    for _ in range(6):
        seg.segment_device(imgs, mode="global")
    seg.ops = Timed()
    for _ in range(10):
        seg.segment_device(imgs, mode="global")
    torch.cuda.synchronize()
    ts = sorted(s.elapsed_time(e) for s, e in ev)
    print("gabor stage inside steps, B=%d %dx%d: median %.4f ms min %.4f ms" % (B, H, W, ts[len(ts) // 2], ts[0]))
    sys.exit(0)
seg = Segmenter()
imgs = torch.from_numpy(synthetic_shard(0, B, H, W)).cuda()
feats = seg.ops.feature_slab(B, H, W)
ts = []
for i in range(12):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); seg.ops.gabor_features(imgs, feats); e.record(); torch.cuda.synchronize()
    ts.append(s.elapsed_time(e))
print("gabor stage B=%d %dx%d: median %.4f ms min %.4f ms" % (B, H, W, sorted(ts)[6], min(ts)))
