#!/bin/bash
# every bench configuration on one box: single graph and the data-parallel form forced on one rank (RCCL group of one)
cd "$GRAFT_REPO_ROOT"
python3 bench.py --steps 100 --warmup 20 > gpurun_out/r4_bench_head.json 2> gpurun_out/r4_bench_head.err
for c in vit_base_con vit_base_adj convvit_base_rec swin_tiny_rec swin_base_rec; do
  python3 bench.py --config $c --steps 100 --warmup 10 --no-cpu-baseline --no-kernel-timing > gpurun_out/r4_bench_$c.json 2> gpurun_out/r4_bench_$c.err
done
for c in vit_base_rec vit_base_con swin_base_rec; do
  python3 bench.py --config $c --force-dist --steps 100 --warmup 10 --no-cpu-baseline --no-kernel-timing > gpurun_out/r4_bench_fd_$c.json 2> gpurun_out/r4_bench_fd_$c.err
done
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r4_bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().split("\n")[-1])
        print(f.split("r4_bench_")[1][:-5].ljust(22), "%.3f ms" % d["ms_per_step"], "%.0f samples/s" % d["value"], "|", d["launch_mode"][:100], "| fallbacks", d.get("eager_fallback_steps"),
              "| tail", d.get("dp_tail_ms_per_rank"))
        if "loader_chain" in d:
            print("   loader_chain", {k: (round(v, 1) if isinstance(v, float) else v) for k, v in d["loader_chain"].items() if k in ("us_per_batch", "wall_us_per_batch", "host_prepare_ms_per_batch", "frac", "error")},
                  "end_to_end", round(d["end_to_end"]["value"]), "roofline frac", round(d["roofline"]["frac"], 4), "voxel", round(d["voxel"]["frac"], 3))
    except Exception as e:
        print(f, "ERR", e, open(f.replace(".json", ".err")).read()[-600:])
PY
