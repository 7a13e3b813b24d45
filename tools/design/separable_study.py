#!/usr/bin/env python3
"""Design study: would two complex 1-D passes beat the 2-D im2col contraction of the Gabor stage?

SPEC.md §2's envelope is isotropic, so in floats every filter factors exactly:
    g(dy, dx) = [e1(dx) exp(i 2 pi f dx cos(theta))] * [e1(dy) exp(i 2 pi f dy sin(theta))] = r(dx) * c(dy),   e1 = the normalised
1-D Gaussian (the 2-D envelope normalised to unit sum IS e1 (x) e1). SURVEY §7.3-1 named this route: 13 + 13 complex taps per
output instead of 169, about 260 instead of 676 int8 MACs once both operands are split into byte digits.

(a) A SPEC-v3 CANDIDATE in exact integers, and how far it is from v2:
      rq, cq = rint(r * 2^15), rint(c * 2^15)                       complex Q15 row / column taps (int16 re, im)
      t[y,x]  = sum_dx rq[dx] * I[y, r(x+dx)]                        complex int32, Q15 grey levels, |t| < 2^23
      u       = (t + 128) >> 8                                       complex int16, Q7, round half up  (the stated rounding)
      v[y,x]  = sum_dy cq[dy] * u[r(y+dy), x]                        complex x complex, int32, Q22
      a       = v >> 15 ;  feature = isqrt(a_re^2 + a_im^2)          the same Q7 magnitude as SPEC.md §3
    on the same octave pyramid with the same reflect border (reflect is separable). Printed: max / RMS deviation from the v2
    features in Q7 units on the three BSD fixtures, and boundary P / R / F of v2 and v3 label maps (same integer Lloyd) on the 24
    packed BSD500 val images, scored by the host mirror of the reference's metrics class (evaluate.metrics, pinned to
    /root/reference/BSD_metrics/metrics.py:58-96 float for float by tests/test_evaluate.py). `--val100 DIR` reads the per-id
    scores of all 100 val ids written by tests/golden/make_bsd_val_scores.py run with GCS_STUDY_V3=1 (the reference's own class).
(b) The per-tile budget of a gfx950 kernel for it - MFMA instructions, vector instructions, LDS bytes for the intermediate with
    its 12 halo rows - priced with the costs tools/ubench/mfma_beside measured (profiles/r3_mfma_beside.txt), beside the same
    budget of the kernel that ships.

    python tools/design/separable_study.py [--val100 DIR] [--skip-val]
"""
import argparse
import json
import math
import os
import sys

import numpy as np
from scipy import ndimage as ndi

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
from oracle import spec_oracle as so, c_oracle as co  # noqa: E402  (a study: the oracle is the v2 reference here)

GOLD = os.path.join(ROOT, "tests", "golden")


# ------------------------------------------------------------------------------------------------ (a) the candidate
def bank_1d(n_scales=4, n_orient=6, ksize=13, f_max=0.4, ratio=math.sqrt(2.0), bandwidth=1.0):
    """Q15 complex row / column taps of every filter: (rq, cq) int64 [F][2][ksize], from SPEC.md §2's parameters."""
    r = (ksize - 1) // 2
    ax = np.arange(-r, r + 1, dtype=np.float64)
    kappa = math.sqrt(math.log(2.0) / 2.0) / math.pi * (2.0 ** bandwidth + 1.0) / (2.0 ** bandwidth - 1.0)
    rq, cq = [], []
    for s in range(n_scales):
        freq = f_max / ratio ** s * 2.0 ** (s // 2)
        sigma = kappa / freq
        e1 = np.exp(-ax * ax / (2.0 * sigma * sigma))
        e1 /= e1.sum()
        for o in range(n_orient):
            th = o * math.pi / n_orient
            pr, pc = 2.0 * math.pi * freq * ax * math.cos(th), 2.0 * math.pi * freq * ax * math.sin(th)
            rq.append(np.rint(np.stack([e1 * np.cos(pr), e1 * np.sin(pr)]) * 2.0 ** 15))
            cq.append(np.rint(np.stack([e1 * np.cos(pc), e1 * np.sin(pc)]) * 2.0 ** 15))
    return np.stack(rq).astype(np.int64), np.stack(cq).astype(np.int64)


def features_v3(img, rq, cq, n_orient):
    """The candidate's canonical (3F, H, W) uint16 features."""
    nf = rq.shape[0]
    h, w = img.shape[:2]
    out = np.empty((3 * nf, h, w), np.uint16)
    n_levels = so.level_of(nf - 1, n_orient) + 1
    for lv, im in enumerate(so.pyramid(img, n_levels)):
        for c in range(3):
            chan = im[:, :, c].astype(np.int64)
            for f in range(nf):
                if so.level_of(f, n_orient) != lv:
                    continue
                t_re = ndi.correlate1d(chan, rq[f, 0], axis=1, mode="reflect")
                t_im = ndi.correlate1d(chan, rq[f, 1], axis=1, mode="reflect")
                u_re, u_im = (t_re + 128) >> 8, (t_im + 128) >> 8
                assert max(np.abs(u_re).max(), np.abs(u_im).max()) < 32768          # the int16 intermediate
                v_re = ndi.correlate1d(u_re, cq[f, 0], axis=0, mode="reflect") - ndi.correlate1d(u_im, cq[f, 1], axis=0, mode="reflect")
                v_im = ndi.correlate1d(u_im, cq[f, 0], axis=0, mode="reflect") + ndi.correlate1d(u_re, cq[f, 1], axis=0, mode="reflect")
                assert max(np.abs(v_re).max(), np.abs(v_im).max()) < 2 ** 31
                a_re, a_im = v_re >> 15, v_im >> 15
                g = so.isqrt_array(a_re * a_re + a_im * a_im).astype(np.uint16)
                out[c * nf + f] = g.repeat(1 << lv, axis=0).repeat(1 << lv, axis=1)[:h, :w]
    return out


def segment_v3(img, k=8, n_iter=10, n_orient=6):
    rq, cq = bank_1d(n_orient=n_orient)
    x = features_v3(img, rq, cq, n_orient).reshape(1, 3 * rq.shape[0], -1)          # (1, D, P): the C oracle's integer Lloyd
    lab, _ = co.kmeans(np.ascontiguousarray(x), k, n_iter)
    return lab.reshape(img.shape[:2])


def fidelity():
    inp = np.load(os.path.join(GOLD, "bsd_inputs.npz"))
    tapq, shift = so.bank()
    rq, cq = bank_1d()
    rows = []
    for i in inp["ids"]:
        img = inp["img_" + str(i)]
        v2 = co.gabor_features(img, tapq.astype(np.int16), shift, 6).astype(np.int64)
        v3 = features_v3(img, rq, cq, 6).astype(np.int64)
        d = v3 - v2
        rows.append((str(i), int(np.abs(d).max()), float(np.sqrt((d * d).mean())), float((d != 0).mean()), float(v2.mean())))
    print("(a) SPEC-v3 candidate vs SPEC v2 features, Q7 units (1 grey level = 128), three BSD fixtures, all 72 planes:")
    print("    id        max |dev|   RMS dev   planes-pixels that differ   mean v2 value")
    for r in rows:
        print("    %-8s  %8d   %7.3f   %22.1f %%   %10.1f" % (r[0], r[1], r[2], 100 * r[3], r[4]))
    # where the deviation comes from: the product of two rounded Q15 taps against the rounded 2-D tap
    prod_re = cq[:, 0, :, None] * rq[:, 0, None, :] - cq[:, 1, :, None] * rq[:, 1, None, :]
    prod_im = cq[:, 0, :, None] * rq[:, 1, None, :] + cq[:, 1, :, None] * rq[:, 0, None, :]
    dq = np.stack([prod_re, prod_im], axis=1) / 2.0 ** 15 - tapq
    print("    tap level: |c(dy) r(dx) / 2^15 - tapq(dy, dx)| max %.2f, RMS %.3f Q15 units (both are roundings of the same float tap)"
          % (np.abs(dq).max(), math.sqrt((dq * dq).mean())))
    return rows


def val_scores(val100):
    from gabor_color_image_segmentation_amd.evaluate import metrics
    from gabor_color_image_segmentation_amd.groundtruth import PackedTruth
    if val100:
        doc = json.load(open(os.path.join(val100, "bsd_val_scores_v3.json")))
        m = doc["mean"]
        print("    BSD500 val, all %d ids, scored by the reference's own metrics class (make_bsd_val_scores.py, GCS_STUDY_V3=1):" % len(doc["ids"]))
        for name in ("v2", "v3"):
            print("      %-3s recall %.4f  precision %.4f  F %.4f" % (name, m[name]["recall"], m[name]["precision"], m[name]["fmeasure"]))
        return m["v2"]["fmeasure"], m["v3"]["fmeasure"], len(doc["ids"])
    pack = np.load(os.path.join(GOLD, "bsd_val_images.npz"))
    pt = PackedTruth(os.path.join(GOLD, "bsd500_truth.npz"))
    ids = [str(i) for i in pack["ids"]]
    f2, f3, worst = [], [], (0.0, None)
    for i in ids:
        img = pack["img_" + i]
        res = {}
        for name, lab in (("v2", pack["labels_" + i]), ("v3", segment_v3(img))):
            m = metrics(img, lab.astype(np.int32), pt[i])
            m.set_boundary_recall()
            m.set_boundary_precision()
            res[name] = 0.0 if m.recall + m.precision == 0 else 2 * m.recall * m.precision / (m.recall + m.precision)
        f2.append(res["v2"])
        f3.append(res["v3"])
        if abs(res["v3"] - res["v2"]) > worst[0]:
            worst = (abs(res["v3"] - res["v2"]), i)
    print("    24 packed BSD500 val images, host mirror of the reference's metrics class: mean F  v2 %.4f   v3 %.4f   (largest"
          " single-image move %.4f on %s)" % (np.mean(f2), np.mean(f3), worst[0], worst[1]))
    return float(np.mean(f2)), float(np.mean(f3)), len(ids)


# ------------------------------------------------------------------------------------------------ (b) the budget
def budget():
    """One workgroup tile = 64 x 32 output pixels of one pyramid level, 3 channels, 12 filters (the level's bank), as the kernel
    that ships: 4 waves, one per SIMD; two workgroups per CU, so a "tile round" is two tiles per CU and two
    wave-tiles per SIMD. Cost model calibrated on the shipped kernel's own counters (profiles/r4_pmc.txt): a
    32x32x32 i8 MFMA holds the pipe 15.9 ns per SIMD (tools/ubench/mfma_beside: 16 - 17 ns), a vector instruction of this
    kernel's mix costs 2.1 ns per SIMD beside the MFMA stream (19.6 us for the 2 x 4 675 a round issues; mfma_beside: 0.7 - 1.2 ns
    for simple ones, 3.2 for v_sqrt_f32, plus the kernel's waits), and the two overlap by 13 %: round = 0.87 (MFMA + VALU)."""
    MF, VA, OVL = 15.9, 2.1, 0.87
    px, ch, nf, ks = 64 * 32, 3, 12, 13
    halo = ks - 1

    def round_us(n_mfma_tile, n_valu_tile):                 # instruction counts per TILE (summed over its 4 waves)
        return OVL * 2 * (n_mfma_tile / 4 * MF + n_valu_tile / 4 * VA) / 1e3

    print("(b) per-tile budget (64 x 32 pixels, 3 channels, the 12 filters of a level; 4 waves, one per SIMD, two tiles per CU):")
    # ---- the kernel that ships: 2-D im2col, K = 224 for 169 taps (7 K-steps), rows = 4 filters x 4 digit-parts x 2 pixel shifts
    mf_now = 21 * (px // 64) * ch                            # 21 MFMAs per 64 pixels and channel
    va_epi = 13 * px * ch * nf // 64                         # 13-instruction exact-magnitude epilogue per output
    va_now = 18700                                           # measured per tile (SQ_INSTS_VALU / tiles): epilogue + staging + addressing
    lds_now = 2 * (32 + halo + 3) * 96 * ch
    r_now = round_us(mf_now, va_now)
    print("    2-D im2col (ships):   %5d MFMA  %6d VALU (%d of them the magnitude epilogue)  LDS %6.1f KB  -> round %.1f us"
          " (measured: 31)" % (mf_now, va_now, va_epi, lds_now / 1024, r_now))
    # ---- separable candidate
    rows_mid = 32 + halo                                     # intermediate rows a tile needs: 44
    n_mid = rows_mid * 64 * ch * nf                          # complex int16 intermediates per tile
    # row pass: A rows = 12 filters x (re, im) x 2 tap digits = 48 -> two 32-row tiles; K = the 13 taps inside one 32-slot step;
    # N = 44 x 64 pixels per channel (every filter shares the window operand, as in the 2-D kernel)
    mf_row = 2 * (rows_mid * 64 // 32) * ch
    # column pass: the operand is the FILTER'S OWN intermediate: nothing is shared across filters. The best MFMA shape is the banded
    # Toeplitz form per (filter, channel): M = 32 output rows x (re, im) x 2 tap digits = 128 rows, K = 44 input rows x (re, im) x
    # 2 intermediate digits = 176 -> 6 steps of 32, N = 64 columns: 4 x 6 x 2 = 48 MFMAs, 13 of 44 band rows non-zero
    mf_col = 4 * 6 * 2 * ch * nf
    # vector work per complex intermediate: round + shift of re and im (4), split into byte digits and pack for the operand (4)
    va_mid = 8 * n_mid // 64
    # per OUTPUT part: P(lo,lo) + 256 (P(lo,hi) + P(hi,lo)) + 65536 P(hi,hi): 5 instructions for re, 5 for im, then the same epilogue
    va_out = 10 * px * ch * nf // 64
    va_sep = va_now + va_mid + va_out
    lds_mid = n_mid * 4                                      # complex int16
    r_sep = round_us(mf_row + mf_col, va_sep)
    print("    separable candidate:  %5d MFMA (%d row + %d column pass)  %6d VALU (+%d for the intermediate, +%d digit combine)  LDS"
          " %6.1f KB  -> round %.1f us" % (mf_row + mf_col, mf_row, mf_col, va_sep, va_mid, va_out, lds_mid / 1024, r_sep))
    print("      the intermediate of one tile (%.0f KB) does not fit the 160 KB of LDS: tiles of 64 x 8 would fit (20 of 20 rows' worth"
          " of row pass per 8 rows of output: 2.5x the row pass, not priced here)" % (lds_mid / 1024))
    print("    int8 MACs per output:  useful  2-D %d, separable %d (row 13 x 2 x 2 + column 13 x 4 x 4);  ISSUED  2-D %d, separable %d:"
          % (169 * 4, 13 * 4 + 13 * 16, mf_now * 32768 // (px * ch * nf), (mf_row + mf_col) * 32768 // (px * ch * nf)))
    print("      the column pass cannot share its operand across filters (in the 2-D form all 12 filters read ONE window operand) and"
          " fills 13 of the 44 rows of its band")
    tiles, slots = 4800 + 1280, 512                          # level-0 + level-1 tiles of 64 BSD images, resident workgroups
    for name, r in (("2-D im2col (ships)", r_now), ("separable candidate", r_sep)):
        print("    predicted stage, 64 images: %-20s %4.1f rounds x %4.1f us + 45 us of pre-passes and strips = %.2f ms"
              % (name, tiles / slots, r, (tiles / slots * r + 45) / 1e3))
    print("    vector-issue floor of the shipped form (all %d VALU of a tile, no MFMA time at all): %.2f ms" % (
        va_now, (tiles / slots * 2 * va_now / 4 * VA + 45e3) / 1e6))
    print("    floor of ANY exact-magnitude form: the epilogue alone, 2 x %d / 4 x 2.1 ns = %.1f us per round -> %.2f ms + 0.045 = %.2f ms"
          % (va_epi, 2 * va_epi / 4 * VA / 1e3, tiles / slots * 2 * va_epi / 4 * VA / 1e6, tiles / slots * 2 * va_epi / 4 * VA / 1e6 + 0.045))
    return r_now, r_sep


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--val100", default=None)
    ap.add_argument("--skip-val", action="store_true")
    args = ap.parse_args()
    fidelity()
    if not args.skip_val:
        f2, f3, n = val_scores(args.val100)
        print("    decision input: |F(v3) - F(v2)| = %.4f on %d images (bar: 0.002)" % (abs(f3 - f2), n))
    t_now, t_sep = budget()
    print("verdict: numerically the candidate is harmless; on this machine it is %.1fx SLOWER per tile round (bar: stage <= 0.30 ms):"
          " not built." % (t_sep / t_now))
