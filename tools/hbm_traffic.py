#!/usr/bin/env python3
"""Per-launch HBM traffic of the two streaming kernels from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE.

Units / corrections (MI355X_MICROARCH.md, HBM section): both counters are in KiB; on gfx950
FETCH_SIZE reports exactly half the bytes of wide (16 B/lane) coalesced streaming reads, so it is
doubled; WRITE_SIZE is exact for 16-B-per-lane streaming stores. Only the batch-64 launches
(the configuration bench.py times) are averaged.
"""
import csv, glob, json, os, sys, collections
out_dir, dst = sys.argv[1], sys.argv[2]
# Round 6: the split Lloyd pass loads 16 + 8 bytes per lane ("other access widths are uncalibrated"): its FETCH_SIZE factor comes from
# a calibration run of the same access pattern on a known byte count (tools/hbm_calibrate.py), passed as a third argument.
calib = json.load(open(sys.argv[3])) if len(sys.argv) > 3 and os.path.exists(sys.argv[3]) else {}
split_factor = calib.get("stream_f<720, 3, 17280>", {}).get("factor")
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out_dir + "/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        key = "gabor_mfma_kernel" if "gabor_mfma_kernel" in n else "kmeans_pass_mfma_kernel" if "kmeans_pass_mfma" in n else \
              "kmeans_pass_native_kernel" if "kmeans_pass_native" in n else \
              "gabor_strip_kernel" if "gabor_strip" in n else \
              "gabor_plane_kernel" if ("gabor_plane" in n or "gabor_down" in n) else None
        if key:
            acc[(key, n.split("(")[0].replace("void ", "").strip(), int(r["Grid_Size"]))][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {}
for (key, full, grid), cs in sorted(acc.items()):
    f = cs.get("FETCH_SIZE", []); w = cs.get("WRITE_SIZE", [])
    factor = split_factor if (split_factor and key == "kmeans_pass_mfma_kernel" and full.rstrip(">").endswith("true")) else 2.0
    res.setdefault(key, []).append(dict(kernel=full, grid_threads=grid, launches=len(f),
        fetch_kib_raw=sum(f) / max(1, len(f)), write_kib=sum(w) / max(1, len(w)), fetch_factor=round(factor, 3),
        hbm_bytes_corrected=int((factor * sum(f) / max(1, len(f)) + sum(w) / max(1, len(w))) * 1024)))
json.dump(res, open(dst, "w"), indent=1)
