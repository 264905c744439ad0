#!/bin/bash
# 400 timed steps of every bench configuration on one box (stability: finite decreasing losses, no eager fall-backs)
cd "$GRAFT_REPO_ROOT"
for c in vit_base_rec vit_base_con vit_base_adj convvit_base_rec swin_tiny_rec swin_base_rec; do
  python3 bench.py --config $c --steps 400 --warmup 10 --no-cpu-baseline --no-kernel-timing > gpurun_out/r4_long_$c.json 2> gpurun_out/r4_long_$c.err
  python3 - "$c" <<'PY'
import json, sys
c = sys.argv[1]
try:
    d = json.loads(open("gpurun_out/r4_long_%s.json" % c).read().strip().split("\n")[-1])
    print(c.ljust(18), "%.3f ms" % d["ms_per_step"], "%.0f/s" % d["value"], "loss %.4f" % d["final_loss"], "fallbacks", d["eager_fallback_steps"], "median %.3f" % d["ms_per_step_median"])
except Exception as e:
    print(c, "ERR", e, open("gpurun_out/r4_long_%s.err" % c).read()[-500:])
PY
done
