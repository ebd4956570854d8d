#!/usr/bin/env python3
"""Per-kernel table (calls per step, average µs, µs per step) from a rocprofv3 --kernel-trace --stats directory."""
import csv
import glob
import os
import sys

src, steps = sys.argv[1], int(sys.argv[2])
files = sorted(glob.glob(f"{src}/**/*_kernel_stats.csv", recursive=True), key=os.path.getmtime)
rows = list(csv.DictReader(open(files[-1])))
tot = 0.0
for r in rows:
    calls, avg = int(r["Calls"]), float(r["AverageNs"]) / 1e3
    per = calls * avg / steps
    tot += per
    print(f"{r['Name'][:100]:100s} calls/step={calls / steps:5.2f} avg_us={avg:10.1f} us/step={per:10.1f}")
print(f"TOTAL kernel us/step = {tot:.1f}")
