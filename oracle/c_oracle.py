"""ctypes wrapper of oracle/gcs_oracle.c — TEST INFRASTRUCTURE ONLY (see spec_oracle.py)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle.so")
_lib = None


def _default_threads():
    """Affinity mask cut down to the cgroup CPU quota: one OpenMP thread per core this process may really use."""
    n = len(os.sched_getaffinity(0))
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:
        pass
    return n


def load():
    global _lib
    if _lib is None:
        subprocess.check_call(["make", "-s", "-C", _HERE, "all"])       # no-op when up to date
        os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")              # idle threads sleep: CPU shares are common
        _lib = C.CDLL(_SO)
        _lib.oracle_set_threads(_default_threads())
    return _lib


def set_threads(n):
    """OpenMP threads of the C oracle (results do not depend on it: integer arithmetic)."""
    load().oracle_set_threads(int(n))


def max_threads():
    return int(load().oracle_max_threads())


def gabor_features(img, tapq, shift, n_orient):
    img = np.ascontiguousarray(img, np.uint8)
    tapq = np.ascontiguousarray(tapq, np.int16)
    h, w = img.shape[:2]
    f, _, ks, _ = tapq.shape
    out = np.empty((3 * f, h, w), np.uint16)
    rc = load().oracle_gabor_features(C.c_void_p(img.ctypes.data), h, w, C.c_void_p(tapq.ctypes.data), f, ks,
                                      int(shift), int(n_orient), C.c_void_p(out.ctypes.data))
    assert rc == 0
    return out


def kmeans(feats, k, n_iter):
    """feats (nimg, D, P) uint16 -> (labels (nimg,P) int32, centroids (k,D) uint16), one codebook."""
    feats = np.ascontiguousarray(feats, np.uint16)
    n, d, p = feats.shape
    labels = np.empty((n, p), np.int32)
    cent = np.empty((k, d), np.uint16)
    rc = load().oracle_kmeans(C.c_void_p(feats.ctypes.data), n, d, C.c_long(p), k, n_iter,
                              C.c_void_p(labels.ctypes.data), C.c_void_p(cent.ctypes.data))
    assert rc == 0
    return labels, cent


def segment_batch(imgs, tapq, shift, n_orient, k=8, n_iter=10, mode="per_image"):
    imgs = np.asarray(imgs)
    b, h, w = imgs.shape[:3]
    feats = np.stack([gabor_features(im, tapq, shift, n_orient) for im in imgs]).reshape(b, -1, h * w)
    if mode == "global":
        return kmeans(feats, k, n_iter)[0].reshape(b, h, w)
    return np.stack([kmeans(feats[i:i + 1], k, n_iter)[0].reshape(h, w) for i in range(b)])
