#!/bin/bash
# one-rank cost of the data-parallel step form, with and without the backward cut (EVP_DP_BACKWARD_CUT), same box, interleaved
cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do
for c in vit_base_rec vit_base_con swin_base_rec; do
  python3 bench.py --config $c --steps 100 --warmup 10 --no-cpu-baseline --no-kernel-timing > gpurun_out/r4_cut_${c}_single_$rep.json 2> gpurun_out/r4_cut_${c}_single_$rep.err
  for cut in 0 1; do
    EVP_DP_BACKWARD_CUT=$cut python3 bench.py --config $c --force-dist --steps 100 --warmup 10 --no-cpu-baseline --no-kernel-timing > gpurun_out/r4_cut_${c}_cut${cut}_$rep.json 2> gpurun_out/r4_cut_${c}_cut${cut}_$rep.err
  done
done
done
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r4_cut_*.json")):
    try:
        d = json.loads(open(f).read().strip().split("\n")[-1])
        print(f.split("r4_cut_")[1][:-5].ljust(30), "%.3f ms" % d["ms_per_step"], "|", d["launch_mode"][:90])
    except Exception as e:
        print(f, "ERR", e, open(f.replace(".json", ".err")).read()[-400:])
PY
