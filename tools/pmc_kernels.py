#!/usr/bin/env python3
"""Per-kernel means of rocprofv3 --pmc counter_collection.csv files: `pmc_kernels.py <dir> [<dir> ...] [--json out.json]`.
Kernel names are cleaned ('(anonymous namespace)::', 'void ', argument lists dropped, template arguments kept), counters of
several passes (one directory per pass) are merged per kernel. Mean is per launch."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def clean(name):
    n = name.replace("(anonymous namespace)::", "").replace("void ", "")
    depth, out = 0, []
    for ch in n:                      # cut at the first '(' outside template brackets = the argument list
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            break
        out.append(ch)
    return re.sub(r"\s+", " ", "".join(out)).strip()[:120]


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    jout = sys.argv[sys.argv.index("--json") + 1] if "--json" in sys.argv else None
    if jout in args:
        args.remove(jout)
    acc, cnt, dur = defaultdict(float), defaultdict(set), defaultdict(list)
    for root in args:
        for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                k = clean(r["Kernel_Name"])
                key = (k, r["Counter_Name"])
                acc[key] += float(r["Counter_Value"])
                cnt[key].add((root, r["Dispatch_Id"]))
                dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)
    table = defaultdict(dict)
    for (k, c), v in acc.items():
        table[k][c] = v / len(cnt[(k, c)])
        table[k]["launches"] = len(cnt[(k, c)])
    for k in table:
        d = sorted(dur[k])
        table[k]["avg_us_profiled"] = sum(d) / len(d)
    for k in sorted(table, key=lambda k: -table[k]["avg_us_profiled"] * table[k]["launches"]):
        t = table[k]
        print("%-110s n=%-5d avg %9.1f us  %s" % (k[:110], t["launches"], t["avg_us_profiled"],
                                                   "  ".join("%s=%.4g" % (c, v) for c, v in sorted(t.items()) if c not in ("launches", "avg_us_profiled"))))
    if jout:
        json.dump(table, open(jout, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
