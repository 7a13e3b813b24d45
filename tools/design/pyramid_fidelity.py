"""Design study (run under /opt/conda/bin/python3.9): how close can an octave pyramid get to
skimage.filters.gabor at full resolution? Float arithmetic, BSD fixture image, all 3 channels' red."""
import math, sys, os
import numpy as np
np.complex = complex
from scipy import ndimage as ndi
from skimage.filters import gabor, gabor_kernel

here = os.path.join(os.path.dirname(__file__), "..", "..", "tests", "golden")
inp = np.load(os.path.join(here, "bsd_inputs.npz"))
img = inp["img_" + str(inp["ids"][0])][:, :, 0].astype(np.float64)
H, W = img.shape
kappa = math.sqrt(math.log(2) / 2) / math.pi * 3.0

def ref_mag(freq, theta):
    re, im = gabor(img, frequency=freq, theta=theta, bandwidth=1.0, mode="reflect")
    return np.hypot(re, im) / np.abs(gabor_kernel(freq, theta=theta, bandwidth=1.0)).sum()   # unit-DC envelope

def taps(freq, theta, R):
    sigma = kappa / freq
    dy, dx = np.mgrid[-R:R + 1, -R:R + 1].astype(float)
    env = np.exp(-(dx * dx + dy * dy) / (2 * sigma * sigma)); env /= env.sum()
    ph = 2 * math.pi * freq * (dx * math.cos(theta) + dy * math.sin(theta))
    return env * np.cos(ph), env * np.sin(ph)

def decimate(x, kern):
    """even-length separable kernel, centre between samples 2i, 2i+1 (block-centred)"""
    n = len(kern); off = n // 2 - 1     # taps cover 2i-off .. 2i+off+1
    k = np.asarray(kern, float) / sum(kern)
    def one(a, axis):
        a = np.moveaxis(a, axis, 0)
        m = a.shape[0] // 2 * 2
        idx = np.arange(0, m, 2)
        out = 0
        for t in range(n):
            j = idx - off + t
            j = np.where(j < 0, -1 - j, j); j = np.where(j >= a.shape[0], 2 * a.shape[0] - 1 - j, j)
            out = out + k[t] * a[j]
        return np.moveaxis(out, 0, axis)
    return one(one(x, 0), 1)

def upsample_nn(x, L, shape):
    y = np.repeat(np.repeat(x, 2 ** L, 0), 2 ** L, 1)
    out = np.zeros(shape); h = min(shape[0], y.shape[0]); w = min(shape[1], y.shape[1])
    out[:h, :w] = y[:h, :w]
    if h < shape[0]: out[h:] = out[h - 1]
    if w < shape[1]: out[:, w:] = out[:, w - 1:w]
    return out

def upsample_lin(x, L, shape):
    s = 2 ** L
    yy = (np.arange(shape[0]) + 0.5) / s - 0.5; xx = (np.arange(shape[1]) + 0.5) / s - 0.5
    return ndi.map_coordinates(x, np.meshgrid(yy, xx, indexing="ij"), order=1, mode="nearest")

def dgain(kern, f):      # 1-D frequency response magnitude of the block-centred kernel at f cycles/px
    n = len(kern); pos = np.arange(n) - (n - 1) / 2
    return abs(np.sum(np.asarray(kern, float) / sum(kern) * np.exp(-2j * math.pi * f * pos)))

KERNS = {"box2": [1, 1], "binom4": [1, 3, 3, 1], "binom6": [1, 5, 10, 10, 5, 1],
         "hb8": [-1, 0, 9, 16, 16, 9, 0, -1]}
NS = int(sys.argv[1]) if len(sys.argv) > 1 else 4
NO = 6
def run(levelfn, R, kern, comp, up):
    pyr = [img]
    for L in range(1, 5): pyr.append(decimate(pyr[-1], KERNS[kern]))
    rows = []
    for s in range(NS):
        freq = 0.4 / math.sqrt(2) ** s
        L = levelfn(s)
        fb = freq * 2 ** L
        errs = []
        for o in range(NO):
            th = o * math.pi / NO
            ref = ref_mag(freq, th)
            tr, ti = taps(fb, th, R)
            re = ndi.correlate(pyr[L], tr, mode="reflect"); im = ndi.correlate(pyr[L], ti, mode="reflect")
            mag = np.hypot(re, im)
            if comp and L:
                g = 1.0
                for l in range(L):
                    g *= dgain(KERNS[kern], freq * 2 ** l * abs(math.cos(th))) * dgain(KERNS[kern], freq * 2 ** l * abs(math.sin(th)))
                mag = mag / g
            mag = (upsample_lin if up == "lin" else upsample_nn)(mag, L, img.shape) if L else mag
            e = mag - ref
            errs.append((np.abs(e).max(), math.sqrt((e * e).mean()), math.sqrt((ref * ref).mean()), ref.max()))
        errs = np.array(errs)
        rows.append((s, L, errs[:, 0].max(), errs[:, 1].max(), errs[:, 2].mean(), errs[:, 3].max()))
    return rows

for name, levelfn, R in (("L=s//2 R7", lambda s: s // 2, 7), ("L=(s-1)//2 R9", lambda s: max(0, (s - 1) // 2), 9),
                         ("L=(s-1)//2 R7", lambda s: max(0, (s - 1) // 2), 7)):
    for kern in ("box2", "binom4", "binom6", "hb8"):
        for comp in (False, True):
            for up in ("nn", "lin"):
                rows = run(levelfn, R, kern, comp, up)
                print(f"{name:16s} {kern:7s} comp={int(comp)} up={up}: " +
                      " | ".join(f"s{s} L{L} max {mx:5.2f} rms {r:5.3f} (ref rms {rr:4.2f} max {rm:4.1f})" for s, L, mx, r, rr, rm in rows if L > 0 or R != 7 and s == 2))
