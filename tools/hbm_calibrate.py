#!/usr/bin/env python3
"""FETCH_SIZE calibration for the split slab's access pattern (MI355X_MICROARCH.md, HBM: "other access widths are uncalibrated:
calibrate on a known byte count in your own access pattern").

    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d DIR -- tools/ubench/read_stream
    python tools/hbm_calibrate.py DIR out.json

tools/ubench/read_stream's variant F reads a known number of bytes the way the split Lloyd pass does (per lane one 16-byte and
one 8-byte nontemporal load from two runs); variant E is the wide pass's pattern (16 bytes per lane), for which the guide's factor
is 2. Prints bytes per reported FETCH_SIZE KiB for both and writes the factors."""
import csv, glob, json, sys, collections
d, dst = sys.argv[1], sys.argv[2]
known = {  # kernel name prefix -> bytes one launch reads (tools/ubench/read_stream.hip)
    "stream_f<720, 3, 17280>": 64 * 607 * 17280,
    "stream_f<720, 4, 17280>": 64 * 607 * 17280,
    "stream_f<1020, 3, 24480>": 64 * 607 * 24480,
    "stream_e<1440, 3>": None,        # launched with two different tile counts: skipped
    "stream_e<1080, 3>": 64 * 607 * 17280,
    "stream_e<1530, 3>": 64 * 607 * 24480,
}
acc = collections.defaultdict(list)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != "FETCH_SIZE":
            continue
        n = r["Kernel_Name"].replace("void ", "").split("(")[0].strip()
        if known.get(n):
            acc[n].append(float(r["Counter_Value"]))
out = {}
for n, v in sorted(acc.items()):
    v = sorted(v)[len(v) // 4:]                         # (the first launches of a kernel find part of the buffer in the Infinity Cache
    kib = sum(v) / len(v)                                #  counted all the same; drop the low quarter against outliers)
    out[n] = dict(launches=len(v), fetch_kib=kib, known_bytes=known[n], factor=known[n] / (kib * 1024))
    print(f"{n:28s} FETCH_SIZE {kib:12.1f} KiB for {known[n]} bytes: factor {out[n]['factor']:.3f}")
json.dump(out, open(dst, "w"), indent=1)
