"""Where the pipelined host path (Segmenter.segment_stream) spends its time per batch: the staging memcpy into pinned memory
(one thread / several), the pinned result allocation, the two PCIe copies alone, and the stream's rate over 10 and 30 batches.

    python tools/host_stream_probe.py            # on the GPU box
"""
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.environ.get("GCS_PKG_ROOT", ROOT))


def med(f, n=9):
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        f()
        ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2] * 1e3


def main():
    import torch
    from gabor_color_image_segmentation_amd.segmenter import Segmenter
    from gabor_color_image_segmentation_amd.synthetic import synthetic_shard
    B, H, W = 64, 321, 481
    imgs = synthetic_shard(0, B, H, W, seed=0)
    px = B * H * W
    print("cpus", len(os.sched_getaffinity(0)), flush=True)
    if "--stream-only" in sys.argv:
        seg = Segmenter(device=torch.device("cuda:0"))
        for dt in (np.int32, np.uint8):
            for _ in seg.segment_stream((imgs for _ in range(6)), mode="global", out_dtype=dt):
                pass
            t0 = time.perf_counter()
            nb = sum(1 for _ in seg.segment_stream((imgs for _ in range(30)), mode="global", out_dtype=dt))
            dt_ = time.perf_counter() - t0
            print("segment_stream %s n=30: %.3f ms per batch, %.0f Mpix/s" % (np.dtype(dt).name, dt_ / nb * 1e3, px * nb / dt_ / 1e6), flush=True)
        return
    pin = torch.empty((B, H, W, 3), dtype=torch.uint8, pin_memory=True)
    pin_np = pin.numpy()
    print("staging memcpy torch copy_  %.3f ms" % med(lambda: pin.copy_(torch.from_numpy(imgs))))
    print("staging memcpy np.copyto    %.3f ms" % med(lambda: np.copyto(pin_np, imgs)))
    for nt in (2, 4, 8):
        pool = ThreadPoolExecutor(nt)
        cuts = [B * i // nt for i in range(nt + 1)]

        def par():
            list(pool.map(lambda i: np.copyto(pin_np[cuts[i]:cuts[i + 1]], imgs[cuts[i]:cuts[i + 1]]), range(nt)))
        print("staging memcpy %d threads    %.3f ms" % (nt, med(par)))
        pool.shutdown()
    for dt in (torch.int32, torch.uint8):
        def alloc():
            return torch.empty((B, H, W), dtype=dt, pin_memory=True)
        print("pinned result alloc %-12s %.3f ms (dropped at once: the caching host allocator hands the block back)" % (dt, med(alloc)))
        keep = []
        t0 = time.perf_counter()
        for _ in range(5):
            keep.append(alloc())
        print("pinned result alloc %-12s %.3f ms each when the caller keeps the results" % (dt, (time.perf_counter() - t0) / 5 * 1e3))
        del keep
    dev = torch.device("cuda:0")
    d_in = torch.empty((B, H, W, 3), dtype=torch.uint8, device=dev)
    d_out = torch.zeros((B, H, W), dtype=torch.int32, device=dev)
    res = torch.empty((B, H, W), dtype=torch.int32, pin_memory=True)

    def h2d():
        d_in.copy_(pin, non_blocking=True)
        torch.cuda.synchronize()

    def d2h():
        res.copy_(d_out, non_blocking=True)
        torch.cuda.synchronize()
    print("H2D 29.6 MB %.3f ms   D2H 39.5 MB %.3f ms" % (med(h2d), med(d2h)))
    seg = Segmenter(device=dev)
    print("torch intra-op threads", torch.get_num_threads(), flush=True)
    seg.segment_batch(imgs, mode="global")
    print("segment_batch host -> host int32 %.3f ms per batch" % med(lambda: seg.segment_batch(imgs, mode="global")))
    seg(imgs[0])
    print("segment(img) host -> host        %.3f ms" % med(lambda: seg(imgs[0]), 21), flush=True)
    for kw in ({}, dict(out_dtype=np.uint8)):
        for _ in seg.segment_stream((imgs for _ in range(3)), mode="global", **kw):
            pass
        for n in (10, 30):
            t0 = time.perf_counter()
            nb = sum(1 for _ in seg.segment_stream((imgs for _ in range(n)), mode="global", **kw))
            dt = time.perf_counter() - t0
            print("segment_stream %s n=%d: %.3f ms per batch, %.0f Mpix/s" % (kw or "int32", n, dt / nb * 1e3, px * nb / dt / 1e6), flush=True)


if __name__ == "__main__":
    main()
