#!/usr/bin/env python3
"""Turn the per-kernel summaries that tools/gpu_round3_profile.sh (round 2: gpu_round2_profile.sh) leaves under gpurun_out/ into the committed profiles/rNN_* files:
kernel stats CSV (copied), pmc_traffic.json (gfx950 correction: reads x 2; the 128x128 / 96x128 forward + data-gradient family as a
launch-weighted mean; the two voxel_bin launches of the verified mode summed per batch), SQ counter summary.
usage: make_profiles.py [r02]"""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out")
P = os.path.join(ROOT, "profiles")


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
    t = json.load(open(os.path.join(G, "pmc_traffic_kernels.json")))
    v = json.load(open(os.path.join(G, "pmc_voxel_kernels.json")))
    sq = json.load(open(os.path.join(G, "pmc_sq_kernels.json")))

    def ent(d, note=None):
        e = dict(fetch_kib_raw=d["FETCH_SIZE"], write_kib=d["WRITE_SIZE"], traffic_bytes=(2 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024,
                 launches_averaged=d["launches"], avg_us_profiled=d["avg_us_profiled"])
        if note:
            e["note"] = note
        return e

    k = {}
    g4 = [x for n, x in t.items() if "gemm_g4_grouped" in n]
    if g4:
        k["gemm_g4_grouped_tn_kernel"] = ent(g4[0], "all weight gradients of the ViT-Base rec step in one launch (256x256 tiles, 4-stage ring of 32-k "
                                                    "stages); r01's ring body moved 11.69 GB per launch")
    fam = [(n, x) for n, x in t.items() if (n.startswith("gemm_kernel<") and ("128, 128" in n or "96, 128" in n)) or n.startswith("g4x_kernel<")]
    L = sum(x["launches"] for _, x in fam)
    f = dict(fetch_kib_raw=sum(x["FETCH_SIZE"] * x["launches"] for _, x in fam) / L, write_kib=sum(x["WRITE_SIZE"] * x["launches"] for _, x in fam) / L,
             launches_averaged=L, note="launch-weighted mean over the %d instantiations of the forward / data-gradient body (128x128 and 96x128 tiles)" % len(fam))
    f["traffic_bytes"] = (2 * f["fetch_kib_raw"] + f["write_kib"]) * 1024
    k["gemm_kernel<"] = f
    for n, x in t.items():
        if n.startswith(("attn_", "ln_", "adamw", "colsum_grouped")):
            k[n] = ent(x)
    for n, x in fam:
        k[n] = ent(x)
    vb = ent(next(x for n, x in v.items() if n.startswith("voxel_bin_kernel<false")))
    vb["per_launch_mean_bytes"] = vb["traffic_bytes"]
    vb["traffic_bytes"] *= 2
    vb["note"] = ("verified mode (assume_sorted=2) = TWO launches of this kernel per batch (pass 1 bins and checks every clip, pass 2 repairs "
                  "flagged clips and is a no-op on sorted input); traffic_bytes is their SUM per 64-clip batch, per_launch_mean_bytes the mean as counted")
    k["voxel_bin_kernel"] = vb
    k["voxel_cuts_kernel"] = ent(v["voxel_cuts_kernel"])
    out = dict(note="rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (counter unit KiB), summarised per kernel by "
                    "tools/pmc_kernels.py on the GPU box. Per MI355X_MICROARCH.md 'HBM': on gfx950 FETCH_SIZE tallies the 128-B requests of "
                    "16-B-per-lane loads at 64 B, so reads are doubled (traffic_bytes = (2 x fetch + write) x 1024); WRITE_SIZE is exact. "
                    "Infinity-Cache hits are included in these fabric-side counters.",
               commands=["tools/gpu_round4_profile.sh (earlier rounds: gpu_round3_profile.sh; rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace --output-format csv -- python3 bench.py "
                         "--no-cpu-baseline --no-kernel-timing --steps 3 --warmup 2; same for tools/voxel_pmc.py)"], kernels=k)
    json.dump(out, open(os.path.join(P, tag + "_pmc_traffic.json"), "w"), indent=1)
    json.dump(dict(note="rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT "
                        "SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace (one pass), per-launch means by tools/pmc_kernels.py. "
                        "SQ_VALU_MFMA_BUSY_CYCLES sums over the 1024 SIMDs; GRBM_GUI_ACTIVE sums over the 8 XCDs.", kernels=sq),
              open(os.path.join(P, tag + "_pmc_sq_kernels.json"), "w"), indent=1)
    shutil.copy(os.path.join(G, "pmc_sq_kernels.txt"), os.path.join(P, tag + "_pmc_sq_kernels.txt"))
    shutil.copy(os.path.join(G, "pmc_traffic_kernels.txt"), os.path.join(P, tag + "_pmc_traffic_kernels.txt"))
    st = "r4" if tag >= "r04" else "r3" if tag >= "r03" else "r2"
    shutil.copy(os.path.join(G, "prof_%s_stats" % st, "%s_kernel_stats.csv" % st), os.path.join(P, tag + "_bench_kernel_stats.csv"))
    for extra, name in (("prof_%s_check.txt" % st, "_roofline_check.txt"), (os.path.join("prof_%s_stats" % st, "%s_memory_copy_stats.csv" % st), "_bench_memory_copy_stats.csv")):
        if st != "r2" and os.path.exists(os.path.join(G, extra)):
            shutil.copy(os.path.join(G, extra), os.path.join(P, tag + name))
    for n in ("gemm_g4_grouped_tn_kernel", "gemm_kernel<", "voxel_bin_kernel"):
        if n in k:
            print(n, round(k[n]["traffic_bytes"] / 1e6, 1), "MB")
    for n, x in sq.items():
        if x.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) > 0:
            print("%-100s %7.1fus mfma_busy %.3f lds_conflict %.2f" % (n[:100], x["avg_us_profiled"], x["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / (x["GRBM_GUI_ACTIVE"] / 8),
                                                                     x["SQ_LDS_BANK_CONFLICT"] / max(x["SQ_LDS_IDX_ACTIVE"], 1)))


if __name__ == "__main__":
    main()
