#!/usr/bin/env python3
"""Kernel-stats CSV (the columns of `rocprofv3 --stats`'s kernel_stats.csv) from a rocpd results .db, for the cases
where rocprofv3 was run with its default database output.  usage: rocpd_stats.py results.db out.csv"""
import csv
import sqlite3
import sys


def main(db, out):
    con = sqlite3.connect(db)
    cur = con.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else "kernel_name"
    start, end = ("start", "end") if "start" in cols else ("start_timestamp", "end_timestamp")
    rows = cur.execute(f"select {name_col}, count(*), sum({end}-{start}), avg({end}-{start}), min({end}-{start}), max({end}-{start}) "
                       f"from kernels group by {name_col} order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    with open(out, "w", newline="") as f:
        w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for n, c, t, a, mn, mx in rows:
            w.writerow([n, c, int(t), round(a, 3), round(100.0 * t / tot, 2), int(mn), int(mx)])
    print(f"{len(rows)} kernels, {tot / 1e6:.3f} ms total -> {out}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
