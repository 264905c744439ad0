#!/usr/bin/env python3
"""Cross-check of bench.py's in-step GEMM timings against rocprofv3: for every kernel instantiation of the bench line's
`gemm_kernels_in_step` table find the rocprofv3 kernel-stats row (a --no-kernel-timing run: averages of the launches inside the
replayed step) and compare avg duration x launches per step; prints the family total both ways and the resulting roofline fractions.
usage: check_roofline.py <bench stdout log> <rocprofv3 kernel_stats.csv>"""
import csv
import json
import re
import sys


def main(bench_log, stats_csv):
    line = [l for l in open(bench_log).read().split("\n") if l.startswith("{")][-1]
    d = json.loads(line)
    rows = list(csv.DictReader(open(stats_csv)))

    def prof_row(kernel):
        # "gemm_kernel<in=bf16,out=bf16,epi=0,transA=0,transB=1,tile=96x128>" -> "gemm_kernel<unsigned short, unsigned short, 0, false, true, 96, 128,"
        m = re.match(r"(gemm|g4x)_kernel<in=(\w+),out=(\w+),epi=(\d),transA=(\d),transB=(\d),tile=(\d+)x(\d+)>", kernel)
        if not m:
            key = kernel.split(" (")[0]
            cands = [r for r in rows if key in r["Name"]]
            return cands[0] if cands else None
        fam, tin, tout, epi, ta, tb, bm, bn = m.groups()
        ty = {"bf16": "unsigned short", "f32": "float"}
        b = {"0": "false", "1": "true"}
        if fam == "gemm":
            pat = "gemm_kernel<%s, %s, %s, %s, %s, %s, %s," % (ty[tin], ty[tout], epi, b[ta], b[tb], bm, bn)
        else:
            pat = "g4x_kernel<true, %s, %d, %d," % ("false" if tb == "1" else "true", int(bm) // 64, int(bn) // 64)
        cands = [r for r in rows if pat in r["Name"] and (fam == "gemm" or (", %s, %s," % (ty[tout], epi)) in r["Name"])]
        return cands[0] if cands else None

    tot_b = tot_p = flops = 0.0
    print("%-78s %5s %10s %10s %7s" % ("kernel", "n", "bench_us", "rocprof_us", "ratio"))
    for k in d.get("gemm_kernels_in_step") or []:
        r = prof_row(k["kernel"])
        pa = float(r["AverageNs"]) / 1e3 if r else float("nan")
        fam = k["kernel"].startswith(("gemm_kernel<", "g4x_kernel<"))
        print("%-78s %5d %10.2f %10.2f %7.3f" % (k["kernel"][:78], k["launches_per_step"], k["avg_us"], pa, k["avg_us"] / pa if r else float("nan")))
        if fam and r:
            tot_b += k["avg_us"] * k["launches_per_step"]
            tot_p += pa * k["launches_per_step"]
            flops += k["tflops"] * 1e12 * k["ms_per_step"] * 1e-3
    peak = d["roofline"]["peak"]
    print("family: bench (in-kernel stamps) %.3f ms -> frac %.4f | rocprofv3 %.3f ms -> frac %.4f | ratio %.3f | bench line roofline.frac %.4f" %
          (tot_b / 1e3, flops / (tot_b * 1e-6) / 1e12 / peak, tot_p / 1e3, flops / (tot_p * 1e-6) / 1e12 / peak, tot_b / tot_p, d["roofline"]["frac"]))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
