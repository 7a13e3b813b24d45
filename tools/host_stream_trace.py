"""Device timeline of the pipelined host path. Two modes:

    rocprofv3 --kernel-trace --memory-copy-trace -d gpurun_out/hst -o hst --output-format csv -- python tools/host_stream_trace.py run
    python tools/host_stream_trace.py show gpurun_out/hst          # prints copies, the kernels they overlap, idle gaps
"""
import csv
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run():
    import numpy as np
    import torch
    from gabor_color_image_segmentation_amd.segmenter import Segmenter
    from gabor_color_image_segmentation_amd.synthetic import synthetic_shard
    imgs = synthetic_shard(0, 64, 321, 481, seed=0)
    seg = Segmenter(device=torch.device("cuda:0"))
    dt = np.uint8 if len(sys.argv) > 2 and sys.argv[2] == "u8" else np.int32
    if len(sys.argv) > 2 and sys.argv[2] == "batch":               # back-to-back segment_batch calls instead of the stream
        import time
        for _ in range(3):
            seg.segment_batch(imgs, mode="global")
        t0 = time.perf_counter()
        for _ in range(8):
            seg.segment_batch(imgs, mode="global")
        print("segment_batch: %.3f ms per call" % ((time.perf_counter() - t0) / 8 * 1e3))
        return
    for _ in seg.segment_stream((imgs for _ in range(3)), mode="global", out_dtype=dt):
        pass
    n = sum(1 for _ in seg.segment_stream((imgs for _ in range(12)), mode="global", out_dtype=dt))
    print("batches", n)


MIN_NS = 30000


def show(d):
    def load(pat):
        rows = []
        for f in glob.glob(os.path.join(d, "**", pat), recursive=True):
            rows += list(csv.DictReader(open(f)))
        return rows
    ker = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:44], r.get("Queue_Id", "?"))
                 for r in load("*kernel_trace.csv"))
    cop = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Direction"].replace("MEMORY_COPY_", ""))
                 for r in load("*memory_copy_trace.csv"))
    gab = [k for k in ker if "gabor_plane" in k[2]]
    tb = gab[-4][0]
    print("kernels %d, SDMA copies %d; timeline from the plane pre-pass of the 4th-last batch (us; kernels of >= 30 us, every copy):"
          % (len(ker), len(cop)))
    ev = [(k[0], k[1], "kernel q%s %s" % (k[3], k[2])) for k in ker if k[0] >= tb] + \
         [(c[0], c[1], "SDMA copy " + c[2]) for c in cop if c[0] >= tb and c[1] - c[0] > 20000]
    run, last = 0, None
    for s, e, name in sorted(ev):
        if e - s < MIN_NS and "copy" not in name.lower():
            continue
        if "kmeans_pass" in name:                      # ten in a row: one line
            run += 1
            last = (s, e)
            continue
        if run:
            print("  ... %d Lloyd passes, the last %9.1f .. %9.1f (%6.1f)" % (run, (last[0] - tb) / 1e3, (last[1] - tb) / 1e3, (last[1] - last[0]) / 1e3))
            run = 0
        print("  %9.1f .. %9.1f  (%7.1f)  %s" % ((s - tb) / 1e3, (e - tb) / 1e3, (e - s) / 1e3, name))
    per = [(gab[i + 1][0] - gab[i][0]) / 1e3 for i in range(len(gab) - 1)]
    print("batch period (plane pre-pass to plane pre-pass, us):", " ".join("%.0f" % p for p in per[-8:]))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run()
    else:
        show(sys.argv[2])
