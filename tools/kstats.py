#!/usr/bin/env python3
"""Print rocprofv3 kernel_stats.csv rows (optionally filtered by substring)."""
import csv, glob, sys
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for f in glob.glob(sys.argv[1] + "/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if pat in r["Name"]:
            print(f"  {r['Name'][:44]:44s} calls={r['Calls']:>4s} avg_us={float(r['AverageNs'])/1e3:9.1f} "
                  f"min={float(r['MinNs'])/1e3:9.1f} max={float(r['MaxNs'])/1e3:9.1f} pct={float(r['Percentage']):5.1f}")
