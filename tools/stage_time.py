#!/usr/bin/env python3
"""Time the stages of the path for any bank: tools/stage_time.py [B] [n_scales] [n_orient] [k] [mode] [H] [W].

Prints the Gabor stage and one Lloyd pass as HIP events measure them INSIDE whole steps, and the step itself (n_iter = 10), on B synthetic
321x481 images, with the algorithmic GB/s of the pass (pyramid-resident bytes) and the int8 TOP/s of the bank.
BASELINE config 4 (8x8 bank, D = 192): tools/stage_time.py 64 8 8."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gabor_color_image_segmentation_amd import Segmenter
from gabor_color_image_segmentation_amd.synthetic import synthetic_shard
a = sys.argv[1:]
B = int(a[0]) if len(a) > 0 else 64
ns = int(a[1]) if len(a) > 1 else 4
no = int(a[2]) if len(a) > 2 else 6
k = int(a[3]) if len(a) > 3 else 8
mode = a[4] if len(a) > 4 else "global"
H = int(a[5]) if len(a) > 5 else 321
W = int(a[6]) if len(a) > 6 else 481
imgs = torch.from_numpy(synthetic_shard(0, B, H, W)).cuda()
seg = Segmenter(n_scales=ns, n_orient=no, k=k)
D = seg.bank.n_features
ws = seg._workspace(B, H, W, mode)
n_sets = B if mode == "per_image" else 1


def ev():
    return torch.cuda.Event(enable_timing=True)


class InStep:
    """HIP events INSIDE whole steps (as bench.py's TimedOps does): the Gabor call of every step and every 3rd Lloyd pass
    (10 passes per step: the sampled position walks through all of them). Stages timed in isolation (this tool up to round 4:
    the same kernel launched back to back) do not add up to the step - an isolated pass repeats on a slab it has just read and
    an isolated Gabor stage starts on an idle chip - so its parts exceeded its step."""

    def __init__(self, ops):
        self._ops, self.g, self.p, self.n = ops, [], [], 0

    def __getattr__(self, name):
        return getattr(self._ops, name)

    def _t(self, lst, fn, *a, **kw):
        s, e = ev(), ev()
        s.record(); fn(*a, **kw); e.record()
        lst.append((s, e))

    def gabor_features(self, *a, **kw):
        return self._t(self.g, self._ops.gabor_features, *a, **kw)

    def _pass(self, fn, *a, **kw):
        self.n += 1
        return self._t(self.p, fn, *a, **kw) if self.n % 3 == 0 else fn(*a, **kw)

    def assign_accumulate(self, *a, **kw):
        return self._pass(self._ops.assign_accumulate, *a, **kw)

    def assign_raster(self, *a, **kw):
        return self._pass(self._ops.assign_raster, *a, **kw)


plain = seg.ops
for _ in range(3):
    seg.segment_device(imgs, mode=mode)
torch.cuda.synchronize()
NSTEP = 12
s0, s1 = ev(), ev()
s0.record()
for _ in range(NSTEP):
    seg.segment_device(imgs, mode=mode)
s1.record()
torch.cuda.synchronize()
ts = s0.elapsed_time(s1) / NSTEP                                               # the step, no events inside
seg.ops = probe = InStep(plain)
for _ in range(NSTEP):
    seg.segment_device(imgs, mode=mode)
torch.cuda.synchronize()
seg.ops = plain
med = lambda l: sorted(a.elapsed_time(b) for a, b in l)[len(l) // 2]
tg, tp = med(probe.g), med(probe.p)
n_iter = seg.n_iter
px = B * H * W
bank = seg.bank
lv = [(3 * min(2, ns - 2 * L) * no, 4 ** L) for L in range(bank.n_levels)]     # (planes, pixel divisor) per pyramid level
feat_b = 2 * sum(d / q for d, q in lv)                                         # feature bytes per full-resolution pixel
ops = sum(2 * bank.ksize ** 2 * (4 * d // 3) * 3 * px / q for d, q in lv)      # int8 MACs x 2
print(f"B={B} {W}x{H} bank {ns}x{no} D={D} ({bank.n_levels} pyramid levels, {feat_b:.0f} feature B/px) k={k} {mode}: "
      f"gabor {tg:.3f} ms ({ops/tg/1e9:.0f} TOP/s int8 two-digit MFMA, exact int32 accumulation"
      f"{' - BASELINE.json names bf16 for the 64-filter bank' if ns * no == 64 else ''}; {(3+feat_b)*px/tg/1e6:.0f} GB/s) | "
      f"pass {tp:.3f} ms ({(feat_b+1)*px/tp/1e6:.0f} GB/s alg) | step {ts:.2f} ms = {px/ts/1e3:.0f} Mpix/s | "
      f"in-step events: gabor + {n_iter} x pass = {tg + n_iter * tp:.2f} ms of the {ts:.2f} ms step (the rest: reduce launches, init)")
