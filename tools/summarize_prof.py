#!/usr/bin/env python3
"""Condense a tools/prof_run.sh output directory into a small text summary for profiles/."""
import collections
import csv
import glob
import sys

src, dst = sys.argv[1], sys.argv[2]
out = []
import os


def newest(pattern):
    """gpurun_out keeps files of earlier runs: take the most recent match only."""
    files = sorted(glob.glob(pattern), key=os.path.getmtime)
    return files[-1:]


for f in newest(f"{src}/trace/*/*_kernel_stats.csv"):
    out.append("# rocprofv3 --kernel-trace --stats (kernel_stats.csv)")
    out += [l.rstrip()[:220] for l in open(f).readlines()[:10]]
for d in ["pmc_sq", "pmc_lds", "pmc_fetch", "pmc_write"]:
    for f in newest(f"{src}/{d}/*/*_counter_collection.csv"):
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        calls = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:70]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        ids = collections.defaultdict(set)
        for r in csv.DictReader(open(f)):
            ids[r["Kernel_Name"][:70]].add(r["Dispatch_Id"])
        out.append(f"# rocprofv3 --pmc pass {d}: counter sums over all dispatches of the kernel (dispatch count in brackets)")
        for k, v in agg.items():
            if "tvae" in k:
                out.append(f"{k} [{len(ids[k])}]")
                out.append("    " + " ".join(f"{a}={b:.0f}" for a, b in sorted(v.items())))
open(dst, "w").write("\n".join(out) + "\n")
print(len(out), "lines")
