#!/usr/bin/env python3
"""Kernel launches (and time) of ONE graph replay from two rocprofv3 --kernel-trace --stats runs of bench.py that differ only in the
number of timed steps: (calls_B - calls_A) / (replays_B - replays_A) per kernel name. usage: per_replay_counts.py A_kernel_stats.csv
B_kernel_stats.csv n_extra_replays"""
import csv
import sys


def load(fn):
    out = {}
    with open(fn) as f:
        for r in csv.DictReader(f):
            out[r["Name"]] = (int(r["Calls"]), float(r["TotalDurationNs"]))
    return out


a, b, n = load(sys.argv[1]), load(sys.argv[2]), float(sys.argv[3])
rows = []
for k in b:
    ca, ta = a.get(k, (0, 0.0))
    cb, tb = b[k]
    if cb != ca:
        rows.append(((tb - ta) / n / 1e3, (cb - ca) / n, k))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print(f"per replay: {sum(r[1] for r in rows):.1f} launches, {tot:.1f} us of kernels")
for us, c, k in rows:
    print(f"{us:9.1f} us {c:7.2f} x  {k[:150]}")
