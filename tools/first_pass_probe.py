#!/usr/bin/env python3
"""Per-pass times of a step (HIP events around every Lloyd pass and the Gabor stage), median over N steps: shows what the
first pass after the Gabor stage costs against the others. Env knobs of the library apply (e.g. GCS_GABOR_PLAIN_TAIL_PCT)."""
import os, sys, statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gabor_color_image_segmentation_amd import Segmenter
from gabor_color_image_segmentation_amd.synthetic import synthetic_shard
B, H, W = 64, 321, 481
N = int(sys.argv[1]) if len(sys.argv) > 1 else 12
imgs = torch.from_numpy(synthetic_shard(0, B, H, W)).cuda()
seg = Segmenter()
ops = seg.ops
out = torch.empty((B, H, W), dtype=torch.int32, device="cuda")
times = {"gabor": [], "step": []}
for i in range(10):
    times[f"pass{i}"] = []
orig_g, orig_a, orig_r = ops.gabor_features, ops.assign_accumulate, ops.assign_raster
cur = {"ev": None, "n": 0}


def ev():
    return torch.cuda.Event(enable_timing=True)


def wrap(name_fn, fn):
    def f(*a, **kw):
        s, e = ev(), ev()
        s.record(); r = fn(*a, **kw); e.record()
        cur["ev"].append((name_fn(), s, e))
        return r
    return f


def pass_name():
    n = cur["n"]; cur["n"] += 1
    return f"pass{n}"


ops.gabor_features = wrap(lambda: "gabor", orig_g)
ops.assign_accumulate = wrap(pass_name, orig_a)
ops.assign_raster = wrap(pass_name, orig_r)
for it in range(N + 8):
    cur["ev"], cur["n"] = [], 0
    s, e = ev(), ev()
    s.record(); seg.segment_device(imgs, mode="global", out=out); e.record()
    torch.cuda.synchronize()
    if it >= 8:
        times["step"].append(s.elapsed_time(e))
        for name, a, b in cur["ev"]:
            times[name].append(a.elapsed_time(b))
print(" ".join(f"{k} {statistics.median(v):.4f}" for k, v in times.items()))
