#!/usr/bin/env python3
"""Reads a rocprofv3 --kernel-trace CSV and reports, for the last `--window-ms` of the trace, how much of the time 0 / 1 / >= 2
kernels were running (do the branches of a captured graph overlap?), and the per-stream / per-queue kernel counts."""
import argparse
import csv
import sys
from collections import Counter

ap = argparse.ArgumentParser()
ap.add_argument("csv")
ap.add_argument("--window-ms", type=float, default=40.0)
ap.add_argument("--anchor", default="adamw", help="the window ends at the end of the last kernel whose name contains this")
ap.add_argument("--dump", default=None, help="write the window's kernels (queue, start us, duration us, name) to this file")
ns = ap.parse_args()
rows = []
with open(ns.csv) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", ""), r.get("Stream_Id", ""), r["Kernel_Name"][:160]))
rows.sort()
anch = [r[1] for r in rows if ns.anchor in r[4]]
t_end = max(anch) if anch else max(r[1] for r in rows)
t0 = t_end - int(ns.window_ms * 1e6)
sel = [r for r in rows if r[0] >= t0 and r[1] <= t_end]
ev = []
for s, e, *_ in sel:
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
depth, last, hist = 0, ev[0][0], Counter()
for t, d in ev:
    hist[min(depth, 3)] += t - last
    last = t
    depth += d
tot = sum(hist.values())
print("kernels in window:", len(sel), " queues:", Counter(r[2] for r in sel).most_common(6), " streams:", Counter(r[3] for r in sel).most_common(6))
for k in sorted(hist):
    print(f"  {k}{'+' if k == 3 else ''} kernels running: {hist[k] / 1e6:8.3f} ms  {100 * hist[k] / tot:5.1f} %")
print(f"  sum of kernel durations {sum(e - s for s, e, *_ in sel) / 1e6:.3f} ms over {tot / 1e6:.3f} ms of wall time")

if ns.dump:
    with open(ns.dump, "w") as f:
        for s_, e_, q, st, name in sel:
            f.write(f"{q:>3s} {(s_ - t0) / 1e3:10.1f} {(e_ - s_) / 1e3:8.1f}  {name}\n")
