#!/bin/bash
# per-kernel split of one bench configuration: tools/gpu_config_prof.sh convvit_base_rec
cd /tmp && export TMPDIR=/tmp
c=${1:-convvit_base_rec}
rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/cfgprof_$c" -- python3 "$GRAFT_REPO_ROOT/bench.py" --config $c --steps 10 --warmup 4 --no-cpu-baseline --no-kernel-timing > "$GRAFT_REPO_ROOT/gpurun_out/cfgprof_$c.log" 2>&1
f=$(find "$GRAFT_REPO_ROOT/gpurun_out/cfgprof_$c" -name "*kernel_stats.csv" | head -1)
cp "$f" "$GRAFT_REPO_ROOT/gpurun_out/cfg_${c}_kernel_stats.csv"
tail -1 "$GRAFT_REPO_ROOT/gpurun_out/cfgprof_$c.log" | cut -c1-200
