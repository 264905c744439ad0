#!/bin/bash
# per-replay kernel census of the headline step (tools/per_replay_counts.py): two kernel-trace runs, 10 and 30 timed steps
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for n in 10 30; do
  rm -rf /tmp/prc_$n
  rocprofv3 --kernel-trace --stats -d /tmp/prc_$n -o t --output-format csv -- python3 $R/bench.py --steps $n --warmup 5 --no-kernel-timing --no-cpu-baseline > $R/gpurun_out/prc_$n.json 2> $R/gpurun_out/prc_$n.err
  cp $(find /tmp/prc_$n -name "*kernel_stats.csv" | head -1) $R/gpurun_out/prc_${n}_kernel_stats.csv
done
python3 $R/tools/per_replay_counts.py $R/gpurun_out/prc_10_kernel_stats.csv $R/gpurun_out/prc_30_kernel_stats.csv 40 > $R/gpurun_out/r4_per_replay.txt
head -50 $R/gpurun_out/r4_per_replay.txt
