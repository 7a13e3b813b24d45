#!/usr/bin/env python3
"""Child process of tests/test_gpu_parity.py::test_a_capture_invalidated_by_another_threads_device_synchronize.

VERDICT r5 item 3: the package no longer issues a device-wide synchronize, but a CALLER's thread (a loader, a logger) still can.
Thread A makes its first call of a new shape - which captures the shape's HIP graph - while thread B calls
torch.cuda.synchronize() in a loop: hipDeviceSynchronize is refused while any thread captures AND invalidates that capture
(profiles/r5_capture_probe.txt). What must hold: A's labels are the oracle's, at most one "graph capture refused" warning, a second
call of the shape is correct too (graph replay or eager launches), other shapes still work afterwards, and the process ends
normally (the invalidated graph is parked, never destroyed: torch 2.10's ~CUDAGraph would std::terminate).

Run in a process of its own so that an abort fails one test instead of ending the test session. Prints one line `OK ...`."""
import os
import sys
import threading
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    import torch
    from gabor_color_image_segmentation_amd import Segmenter
    from gabor_color_image_segmentation_amd import segmenter as sg
    from gabor_color_image_segmentation_amd.synthetic import synthetic_batch
    from oracle import spec_oracle as so

    seg = Segmenter(n_iter=3)
    seg(synthetic_batch(1, 40, 56, seed=1)[0])              # library, side stream, allocator: warmed up on another shape
    torch.cuda.current_stream().synchronize()

    img = synthetic_batch(1, 56, 88, seed=97)[0]
    want = so.segment(img, n_iter=3)
    stop, started = threading.Event(), threading.Event()
    stats = dict(ok=0, refused=0)

    def hammer():                                            # thread B: a device-wide synchronize, again and again
        while not stop.is_set():
            try:
                torch.cuda.synchronize()
                stats["ok"] += 1
            except RuntimeError:                             # "operation not permitted when stream is capturing"
                stats["refused"] += 1
            started.set()

    t = threading.Thread(target=hammer)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        t.start()
        started.wait(10)
        try:
            first = seg(img)                                 # thread A: first call of the shape = eager step + capture
        finally:
            stop.set()
            t.join(60)
        second = seg(img)
    n_warn = sum("graph capture refused" in str(x.message) for x in w)
    ent = seg._graphs[(1, 56, 88, "per_image", np.dtype(np.int32).str)]
    assert np.array_equal(first, want), "first call differs from the oracle"
    assert np.array_equal(second, want), "second call differs from the oracle"
    assert n_warn <= 1
    assert (ent["graph"] is None) == (n_warn == 1), "a refused capture must warn, a taken one must not"
    other = synthetic_batch(1, 48, 72, seed=5)[0]            # the stream and the guard are usable afterwards: a new shape captures
    assert np.array_equal(seg(other), so.segment(other, n_iter=3))
    torch.cuda.synchronize()
    print(f"OK capture {'invalidated -> eager launches' if n_warn else 'taken'}; thread B: {stats['ok']} synchronizes ok, "
          f"{stats['refused']} refused; parked graphs {len(sg._CAPTURES._failed)}", flush=True)


if __name__ == "__main__":
    main()
