"""BSD500 `val` split, boundary recall / precision / F — the quality gate of the Gabor + k-means slot.

Run under /opt/conda/bin/python3.9 with cwd = /root/reference/BSD_metrics (this container only):

    cd /root/reference/BSD_metrics && PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg \
        /opt/conda/bin/python3.9 -W ignore /root/repo/tests/golden/make_bsd_val_scores.py /root/repo/tests/golden [n_workers]

For an explicit SORTED id list (the 100 ids of data/Berkeley/val) the loop of BSD_metrics/script.py:19-38 is run with
four occupants of the slot at script.py:30, each scored by the reference's OWN `metrics` class
(metrics.py:58-74 recall, :77-96 precision; F = 2PR/(P+R) is SURVEY.md §8 a11):

  v2     SPEC.md as it stands (octave pyramid, 13x13 Q15 taps, integer Lloyd): oracle/gcs_oracle.c. The HIP path is
         bit-identical to this oracle, so these ARE the product's scores (tests/test_gpu_golden.py checks `==`).
  r1     round 1's SPEC: every scale at full resolution inside one 15x15 frame, taps at the largest exponent that
         fits (Q18), same integer Lloyd schedule. Same C oracle, driven with that bank and one pyramid level.
  float  float64, full resolution: |skimage.filters.gabor(channel, frequency=f_s, theta, bandwidth=1)| for the same
         4 x 6 (frequency, orientation) grid (skimage picks the support: +-ceil(3 sigma)), then the SAME Lloyd schedule
         in float64 (init pixels of SPEC.md §4, 10 passes, mean update, lowest index on ties).
  slic   skimage.segmentation.slic(img, n_segments=300, compactness=10.0): what the reference ships in the slot.

Only data leaves the reference tree (decoded arrays, numbers); no reference source is copied. Writes
  bsd_val_scores.json   per-id and mean P / R / F of the four occupants (+ regions, + the remaining metrics of
                        get_metrics() for v2)
  bsd_val_images.npz    decoded uint8 images of the first N_PACK ids (the GPU box has no JPEG decoder input)
"""
import json
import math
import os
import sys
import time
from multiprocessing import Pool

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
N_PACK = 24
K, N_ITER, N_SCALES, N_ORIENT = 8, 10, 4, 6
F_MAX, RATIO = 0.4, math.sqrt(2.0)
STUDY_V3 = bool(os.environ.get('GCS_STUDY_V3'))       # see one_image


def r1_bank():
    """Round 1's SPEC.md §2 (git 31fbdbc): full-resolution frequencies f_max / ratio**s in a 15x15 frame."""
    ks, r = 15, 7
    ax = np.arange(-r, r + 1, dtype=np.float64)
    dy = ax[:, None] * np.ones((1, ks))
    dx = np.ones((ks, 1)) * ax[None, :]
    kappa = math.sqrt(math.log(2.0) / 2.0) / math.pi * 3.0
    taps = []
    for s in range(N_SCALES):
        freq = F_MAX / RATIO ** s
        sigma = kappa / freq
        env = np.exp(-(dx * dx + dy * dy) / (2.0 * sigma * sigma))
        env /= env.sum()
        for o in range(N_ORIENT):
            th = o * math.pi / N_ORIENT
            ph = 2.0 * math.pi * freq * (dx * math.cos(th) + dy * math.sin(th))
            taps.append(np.stack([env * np.cos(ph), env * np.sin(ph)]))
    taps = np.stack(taps)
    e = int(math.floor(math.log2(32639.0 / np.abs(taps).max())))
    return np.rint(taps * 2.0 ** e).astype(np.int16), e - 7


def float_lloyd(x):
    """SPEC.md §4's schedule on float64 features x (P, D)."""
    p = x.shape[0]
    c = x[[((2 * j + 1) * p) // (2 * K) for j in range(K)]].copy()
    x2 = (x * x).sum(axis=1)
    lab = None
    for t in range(N_ITER):
        d = x2[:, None] - 2.0 * (x @ c.T) + (c * c).sum(axis=1)[None, :]
        lab = d.argmin(axis=1)                       # first minimum = lowest index
        if t < N_ITER - 1:
            for j in range(K):
                m = lab == j
                if m.any():
                    c[j] = x[m].mean(axis=0)
    return lab


def one_image(i):
    sys.path.insert(0, '.')
    sys.path.insert(0, REPO)
    np.complex = complex      # skimage 0.18 still spells the dtype with the alias numpy 1.24 removed
    from skimage.io import imread
    from skimage.filters import gabor
    from skimage.segmentation import slic
    from groundtruth import get_segment_from_filename
    from metrics import metrics
    from oracle import c_oracle, spec_oracle

    c_oracle.set_threads(1)
    t0 = time.time()
    img = imread('data/Berkeley/val/' + i + '.jpg')
    segs = get_segment_from_filename(i)
    assert img.dtype == np.uint8 and img.ndim == 3 and len(segs) > 0
    h, w = img.shape[:2]
    maps = {}
    tapq, shift = spec_oracle.bank()
    maps['v2'] = c_oracle.segment_batch(img[None], tapq.astype(np.int16), shift, N_ORIENT, K, N_ITER)[0]
    if STUDY_V3:
        # design study only (tools/design/separable_study.py): the separable SPEC-v3 CANDIDATE beside v2, nothing else; the
        # result goes to bsd_val_scores_v3.json, the goldens are not touched
        sys.path.insert(0, os.path.join(REPO, 'tools', 'design'))
        import separable_study
        maps['v3'] = separable_study.segment_v3(img, K, N_ITER, N_ORIENT)
        res = {}
        for name, lab in maps.items():
            m = metrics(img, lab.astype(np.int32), segs)
            m.set_boundary_recall()
            m.set_boundary_precision()
            r, p = float(m.recall), float(m.precision)
            res[name] = {'regions': float(m.n_segments), 'recall': r, 'precision': p,
                         'fmeasure': 0.0 if r + p == 0 else 2.0 * p * r / (p + r)}
        print(i, ' '.join('%s F=%.4f' % (n, res[n]['fmeasure']) for n in res), '%.0fs' % (time.time() - t0), flush=True)
        return i, img, res, maps['v2'].astype(np.uint8)
    tq1, sh1 = r1_bank()
    # n_orient = F puts every filter on pyramid level 0: round 1's full-resolution bank
    maps['r1'] = c_oracle.segment_batch(img[None], tq1, sh1, tq1.shape[0], K, N_ITER)[0]
    feats = []
    for c in range(3):
        chan = img[:, :, c].astype(np.float64)
        for s in range(N_SCALES):
            for o in range(N_ORIENT):
                re, im = gabor(chan, frequency=F_MAX / RATIO ** s, theta=o * math.pi / N_ORIENT, bandwidth=1.0)
                feats.append(np.hypot(re, im).ravel())
    maps['float'] = float_lloyd(np.stack(feats, axis=1)).reshape(h, w)
    maps['slic'] = slic(img, n_segments=300, compactness=10.0)
    res = {}
    for name, lab in maps.items():
        m = metrics(img, lab.astype(np.int32), segs)
        if name == 'v2':
            m.set_metrics()
            res[name] = {k: float(v) for k, v in m.get_metrics().items()}
        else:
            m.set_boundary_recall()
            m.set_boundary_precision()
            res[name] = {'regions': float(m.n_segments), 'recall': float(m.recall), 'precision': float(m.precision)}
        r, p = res[name]['recall'], res[name]['precision']
        res[name]['fmeasure'] = 0.0 if r + p == 0 else 2.0 * p * r / (p + r)
    print(i, img.shape, ' '.join('%s F=%.4f' % (n, res[n]['fmeasure']) for n in res), '%.0fs' % (time.time() - t0), flush=True)
    return i, img, res, maps['v2'].astype(np.uint8)


def main():
    out_dir = sys.argv[1]
    workers = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    ids = sorted(n[:-4] for n in os.listdir('data/Berkeley/val') if n.endswith('.jpg'))
    if len(sys.argv) > 3:
        ids = ids[:int(sys.argv[3])]
    with Pool(workers) as pool:
        rows = pool.map(one_image, ids, chunksize=1)
    per_id = {i: res for i, _, res, _ in rows}
    means = {}
    if STUDY_V3:
        for name in ('v2', 'v3'):
            means[name] = {k: float(np.mean([per_id[i][name][k] for i in ids])) for k in ('recall', 'precision', 'fmeasure', 'regions')}
        json.dump({'split': 'val', 'ids': ids, 'mean': means, 'per_id': per_id,
                   'scored_by': 'BSD_metrics/metrics.py:58-96 (the reference class)'},
                  open(os.path.join(out_dir, 'bsd_val_scores_v3.json'), 'w'), indent=1, sort_keys=True)
        for name in means:
            print(name, means[name])
        return
    for name in ('v2', 'r1', 'float', 'slic'):
        means[name] = {k: float(np.mean([per_id[i][name][k] for i in ids])) for k in ('recall', 'precision', 'fmeasure', 'regions')}
        # F of the dataset-mean P and R, beside the mean of the per-image F
        r, p = means[name]['recall'], means[name]['precision']
        means[name]['f_of_means'] = 2.0 * p * r / (p + r)
    doc = {
        'split': 'val', 'ids': ids, 'k': K, 'n_iter': N_ITER, 'bank': '%dx%d' % (N_SCALES, N_ORIENT),
        'scored_by': 'BSD_metrics/metrics.py:58-96 (the reference class, scikit-image 0.18.3), F = 2PR/(P+R)',
        'occupants': {'v2': 'SPEC.md (octave pyramid, 13x13, Q15) - oracle/gcs_oracle.c == the HIP path',
                      'r1': "round 1's SPEC: full resolution, 15x15 frame, Q18 taps, same integer Lloyd",
                      'float': 'float64 skimage.filters.gabor at full resolution + float Lloyd, same schedule',
                      'slic': 'skimage.segmentation.slic(n_segments=300, compactness=10.0), script.py:30'},
        'mean': means, 'per_id': per_id,
    }
    json.dump(doc, open(os.path.join(out_dir, 'bsd_val_scores.json'), 'w'), indent=1, sort_keys=True)
    pack = {}
    for i, img, _, lab in rows[:N_PACK]:
        pack['img_' + i] = img
        pack['labels_' + i] = lab
    np.savez_compressed(os.path.join(out_dir, 'bsd_val_images.npz'), ids=np.array(ids[:N_PACK]), **pack)
    for name in means:
        print(name, means[name])


if __name__ == '__main__':
    main()
