#!/usr/bin/env python3
"""Print the per-kernel time table of a rocprofv3 --kernel-trace --stats run: scripts/kernel_breakdown.py <dir> [steps]"""
import csv, glob, os, sys
d = sys.argv[1]
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
f = sorted(glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime)[-1]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:36]:
    ns = float(r["TotalDurationNs"])
    print(f"{r['Name'][:72]:72s} {int(r['Calls']):5d} {ns / 1e6 / steps:8.3f} ms/step {ns / tot * 100:5.1f}% avg {float(r['AverageNs']) / 1e3:8.1f} us")
print(f"total {tot / 1e6 / steps:.3f} ms/step over {steps} steps")
