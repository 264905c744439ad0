#!/usr/bin/env python3
"""Sums a rocprofv3 --pmc counter_collection.csv per kernel: `pmc_sum.py <dir> [substring]` prints
kernel, counter, launches, mean value per launch."""
import csv, glob, os, sys
from collections import defaultdict
root, want = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
acc, cnt = defaultdict(float), defaultdict(set)
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if want not in k:
            continue
        key = (k.split("(")[0][-60:], r["Counter_Name"])
        acc[key] += float(r["Counter_Value"])
        cnt[key].add(r["Dispatch_Id"])
for key in sorted(acc):
    n = len(cnt[key])
    print("%-60s %-12s launches=%d mean=%.3f" % (key[0], key[1], n, acc[key] / n))
