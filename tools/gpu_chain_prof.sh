#!/bin/bash
# per-kernel split of the loader chain (10 batches, eager launches: the captured form replays the same kernels)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/chainprof" -- python3 "$GRAFT_REPO_ROOT/tools/loader_chain_prof.py" > "$GRAFT_REPO_ROOT/gpurun_out/chainprof.log" 2>&1
f=$(find "$GRAFT_REPO_ROOT/gpurun_out/chainprof" -name "*kernel_stats.csv" | head -1)
cp "$f" "$GRAFT_REPO_ROOT/gpurun_out/chain_kernel_stats.csv"
cut -c1-60 "$f" | head -3; awk -F'","' '{print substr($1,2,70), $2, $4}' "$f" | head -14
