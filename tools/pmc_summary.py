#!/usr/bin/env python3
"""Average each PMC counter per kernel name over the dispatches of a rocprofv3 --pmc run."""
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0][-64:]
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in sorted(acc.items()):
    if not any(x in k for x in ("gabor", "kmeans_pass", "kmeans_assign")):
        continue
    print(k)
    for c, v in sorted(cs.items()):
        print(f"   {c:32s} n={len(v):4d} mean={sum(v)/len(v):16.1f}")
